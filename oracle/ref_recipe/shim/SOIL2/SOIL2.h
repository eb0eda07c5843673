/* TEST INFRASTRUCTURE (oracle/_ref build only): P3/main.cpp:16 includes <SOIL2/SOIL2.h> but calls
 * nothing from it; an empty stand-in lets the file compile. */
