/* TEST INFRASTRUCTURE (oracle/_ref build only): see glew.h in this directory. */
#include "glew.h"
