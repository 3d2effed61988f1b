// Error reporting + version for the C-ABI library (include/cldrd_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int cldrd_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return 1;
}

extern "C" const char* cldrd_last_error(void) { return g_err; }
extern "C" int cldrd_version(void) { return 100; }
extern "C" int cldrd_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
