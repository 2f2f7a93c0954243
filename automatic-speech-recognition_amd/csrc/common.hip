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

// compute units of the current device (init-once attribute cache; partitioned / CU-masked devices report fewer than 256)
int las_device_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    return cus;
}
