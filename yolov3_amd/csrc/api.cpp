// Error plumbing + version for libyolov3_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/yolov3_hip.h"

static thread_local char g_err[512] = "";

void y3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y3_last_error(void) { return g_err; }
extern "C" int y3_abi_version(void) { return Y3_ABI_VERSION; }
