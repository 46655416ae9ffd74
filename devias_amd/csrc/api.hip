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

extern "C" int devias_version(void) { return 130; }   // 110: multi-tensor optimizer entry points, 120: devias_fame_*, 130: counters + options

// ---- launch counters: which kernel family served a call (tests assert that the measured kernels are the ones under test) ----
#include <atomic>
static std::atomic<int64_t> g_cnt[DEVIAS_CNT_MAX];
void devias_count(int id) { if (id >= 0 && id < DEVIAS_CNT_MAX) g_cnt[id].fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t devias_counter(int32_t id) { return (id >= 0 && id < DEVIAS_CNT_MAX) ? g_cnt[id].load(std::memory_order_relaxed) : -1; }
extern "C" void devias_counters_reset(void) { for (int i = 0; i < DEVIAS_CNT_MAX; ++i) g_cnt[i].store(0, std::memory_order_relaxed); }

extern "C" int devias_set_option(const char* name, int32_t value) {
    if (!name) return devias_set_error(DEVIAS_EINVAL, "devias_set_option: null name");
    if (devias_gemm_set_option(name, value) || devias_attn_set_option(name, value)) return DEVIAS_OK;
    return devias_set_error(DEVIAS_EINVAL, "devias_set_option: unknown option '%s'", name);
}

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
