// libapx.so core: version, error string, device query.
#include "apx_common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void apx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int apx_version(void) { return 100; }
extern "C" const char* apx_last_error(void) { return g_err; }
extern "C" int apx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
