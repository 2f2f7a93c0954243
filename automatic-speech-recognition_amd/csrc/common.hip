// common.hip -- version / error plumbing of the C ABI.
#include "las_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void las_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int las_version(void) { return 100; }  // 0.1.0
extern "C" const char* las_last_error(void) { return g_err; }
