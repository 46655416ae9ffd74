// Library-level entry points: version, thread-local error string, device info.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int devias_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int devias_version(void) { return 120; }   // 0.1.20: + multi-tensor optimizer entry points (110), + devias_fame_* (120)

extern "C" const char* devias_last_error(void) { return g_err; }

extern "C" int devias_device_info(int device, int64_t* out5) {
    if (!out5) return devias_set_error(DEVIAS_EINVAL, "devias_device_info: null output");
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return devias_set_error(DEVIAS_ELAUNCH, "devias_device_info: %s", hipGetErrorString(e));
    out5[0] = p.multiProcessorCount;
    out5[1] = p.clockRate;
    out5[2] = (int64_t)p.sharedMemPerBlock;
    out5[3] = p.warpSize;
    int arch = 0;
    const char* s = strstr(p.gcnArchName, "gfx");
    if (s) arch = atoi(s + 3);
    out5[4] = arch;
    return DEVIAS_OK;
}
