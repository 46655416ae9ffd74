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

extern "C" int devias_version(void) { return 167; }   // 167: devias_mhsa_*_flags (DEVIAS_ATTN_Q_PRESCALED), devias_block_args grew by WqkvS / qkv_biasS (recompile callers), option attn_qpre; 166: devias_block_args grew by the optional transposed weight copies W*T (recompile callers); 165: launch counters per dK / dV kernel form (DEVIAS_CNT_DKDV*, DEVIAS_CNT_MAX 24), option gemm_epi_spec (additive); 164: devias_mhsa_bwd uses its ws again (row statistics of the one-wave-per-SIMD dK / dV kernel), devias_mhsa_bwd_bias accepts dbv = NULL where devias_mhsa_bwd_bias_dv_from_do() says so; 163: devias_get_option, devias_gemm_release_queue_stream (additive); 162: devias_mhsa_bwd_bias (additive); 161: devias_mhsa_fwd_dropout / _bwd_dropout (additive); 160: devias_loss_dims.scene_ce (struct grew); 110: multi-tensor optimizer entry points, 120: devias_fame_*, 130: counters + options, 140: stream-K GEMM (args struct grew), 150: fused regions, roctx ranges

// ---- launch counters: which kernel family served a call (tests assert that the measured kernels are the ones under test) ----
#include <atomic>
static std::atomic<int64_t> g_cnt[DEVIAS_CNT_MAX];
void devias_count(int id) { if (id >= 0 && id < DEVIAS_CNT_MAX) g_cnt[id].fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t devias_counter(int32_t id) { return (id >= 0 && id < DEVIAS_CNT_MAX) ? g_cnt[id].load(std::memory_order_relaxed) : -1; }
extern "C" void devias_counters_reset(void) { for (int i = 0; i < DEVIAS_CNT_MAX; ++i) g_cnt[i].store(0, std::memory_order_relaxed); }

extern "C" int devias_set_option(const char* name, int32_t value) {
    if (!name) return devias_set_error(DEVIAS_EINVAL, "devias_set_option: null name");
    if (devias_gemm_set_option(name, value) || devias_attn_set_option(name, value)) return DEVIAS_OK;
    if (!strcmp(name, "regions_defer")) { devias_defer_enabled() = value; return DEVIAS_OK; }
    return devias_set_error(DEVIAS_EINVAL, "devias_set_option: unknown option '%s'", name);
}

extern "C" int devias_get_option(const char* name, int32_t* value) {
    if (!name || !value) return devias_set_error(DEVIAS_EINVAL, "devias_get_option: null argument");
    int v = 0;
    if (devias_gemm_get_option(name, &v) || devias_attn_get_option(name, &v)) { *value = v; return DEVIAS_OK; }
    if (!strcmp(name, "regions_defer")) { *value = devias_defer_enabled(); return DEVIAS_OK; }
    return devias_set_error(DEVIAS_EINVAL, "devias_get_option: unknown option '%s'", name);
}

extern "C" const char* devias_last_error(void) { return g_err; }
DeviasDeferList*& devias_defer_slot() { static thread_local DeviasDeferList* slot = nullptr; return slot; }
int& devias_defer_enabled() { static int on = [] { const char* e = getenv("DEVIAS_REGIONS_DEFER"); return e ? atoi(e) : 1; }(); return on; }

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


// ---- gradient-bucket all-reduce for hosts that own an RCCL communicator (SURVEY.md §8b minimum export set) ---------------------------------
// The Python host of this repository reaches RCCL through torch.distributed (devias_amd/parallel.py); a C++ host passes its ncclComm_t here.
// librccl is resolved at the first call (dlsym on the already-loaded image first -- inside a PyTorch process that is torch's bundled
// librccl.so --, then dlopen("librccl.so")), so the library itself keeps its single link dependency on the HIP runtime.
#include <dlfcn.h>
typedef int (*devias_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static devias_nccl_allreduce_fn resolve_allreduce() {
    static devias_nccl_allreduce_fn fn = [] {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!sym) {
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) sym = dlsym(h, "ncclAllReduce");
        }
        return (devias_nccl_allreduce_fn)sym;
    }();
    return fn;
}
extern "C" int devias_allreduce_bucket(void* nccl_comm, void* bucket, int64_t count, int32_t dtype, void* stream) {
    if (!nccl_comm || !bucket || count <= 0) return devias_set_error(DEVIAS_EINVAL, "devias_allreduce_bucket: null communicator / bucket or empty bucket");
    if (dtype != DEVIAS_F32 && dtype != DEVIAS_BF16) return devias_set_error(DEVIAS_EINVAL, "devias_allreduce_bucket: bad dtype %d", dtype);
    devias_nccl_allreduce_fn fn = resolve_allreduce();
    if (!fn) return devias_set_error(DEVIAS_EUNSUPPORTED, "devias_allreduce_bucket: ncclAllReduce not found (librccl.so is not loadable)");
    const int nccl_dtype = dtype == DEVIAS_F32 ? 7 /* ncclFloat32 */ : 9 /* ncclBfloat16 */;
    const int rc = fn(bucket, bucket, (size_t)count, nccl_dtype, 0 /* ncclSum */, nccl_comm, (hipStream_t)stream);
    if (rc != 0) return devias_set_error(DEVIAS_ELAUNCH, "devias_allreduce_bucket: ncclAllReduce returned %d", rc);
    return DEVIAS_OK;
}

// nothing persistent is allocated by the library; shutdown resets the launch counters (kept for hosts written against SURVEY.md §8b's export list)
extern "C" void devias_shutdown(void) { devias_counters_reset(); }


// ---- ROCTX ranges (roctx_shim.h) ------------------------------------------------------------------------------------------------------------
#include "roctx_shim.h"
#include <stdlib.h>
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static int g_roctx_state = -1;          // -1 = not resolved yet, 0 = off, 1 = on
static int roctx_resolve() {
    const char* e = getenv("DEVIAS_ROCTX");
    if (!e || atoi(e) == 0) return 0;
    const char* libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
    for (const char* l : libs) {
        void* h = dlopen(l, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return 1;
    }
    return 0;
}
bool devias_roctx_enabled(void) {
    if (g_roctx_state < 0) g_roctx_state = roctx_resolve();
    return g_roctx_state == 1;
}
void devias_roctx_push(const char* name) { if (devias_roctx_enabled()) g_roctx_push(name); }
void devias_roctx_pop(void) { if (devias_roctx_enabled()) g_roctx_pop(); }
// for hosts (the Python engine marks the loss, the optimizer and whole steps with these)
extern "C" void devias_range_push(const char* name) { devias_roctx_push(name ? name : "devias"); }
extern "C" void devias_range_pop(void) { devias_roctx_pop(); }
