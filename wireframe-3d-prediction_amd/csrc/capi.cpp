// Host-side glue shared by every entry point of libwf3d.so: version + the
// thread-local error string behind wf3d_last_error().
#include <stdarg.h>
#include <stdio.h>

#include "wf3d_common.h"

static thread_local char g_err[512] = "";

void wf3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int wf3d_version(void) { return WF3D_VERSION; }
extern "C" const char* wf3d_last_error(void) { return g_err; }
