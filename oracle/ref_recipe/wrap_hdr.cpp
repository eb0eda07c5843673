/* TEST INFRASTRUCTURE -- oracle/_ref: the reference's lib/hdrloader.cpp (EZRT_REF_HDRLOAD, passed
 * by build_ref.py) compiled from where it lies, as its own translation unit (hdrloader.h has no
 * include guard).  The one source-level patch is the one VERDICT r1 names: sscanf("%ld") into
 * `int` (lib/hdrloader.cpp:68, undefined behaviour on LP64) is redirected to "%d" by a macro. */
#include <stdio.h>
#define sscanf(buf, fmt, a, b) sscanf(buf, "-Y %d +X %d", a, b)
#include EZRT_REF_HDRLOAD
