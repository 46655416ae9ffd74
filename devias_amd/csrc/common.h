// Shared device/host helpers for libdevias_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/devias_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

int devias_set_error(int code, const char* fmt, ...);
int devias_policy_gemm_wt(void);                            // option gemm_wt (gemm.hip): the fused encoder block may run its dgrad GEMMs on transposed weight copies
void devias_count(int id);                                  // launch counters (api.hip): DEVIAS_CNT_* of include/devias_amd.h
int devias_gemm_set_option(const char* name, int value);    // per-module option handlers behind devias_set_option: 1 = name known
int devias_attn_set_option(const char* name, int value);
int devias_gemm_get_option(const char* name, int* value);    // ... and behind devias_get_option
int devias_attn_get_option(const char* name, int* value);
extern "C" int32_t devias_policy_gemm_cus(void);             // gemm.hip: CUs the big-tile grids count on (device CUs - option gemm_reserve_cus)

#define DEVIAS_CHECK_LAUNCH(name)                                                              \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess)                                                                 \
            return devias_set_error(DEVIAS_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define DEVIAS_REQUIRE(cond, ...)                                       \
    do {                                                                \
        if (!(cond)) return devias_set_error(DEVIAS_EINVAL, __VA_ARGS__); \
    } while (0)

// compute units of the CURRENT device (cached per device id: a process that calls a size query before torch.cuda.set_device(local_rank) must not
// pin device 0's count for the device it later runs on -- ADVICE r3)
static inline int devias_device_cus() {
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev];
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev] = v;
    }
    return v;
}

// ---- deferred final reductions ---------------------------------------------------------------------------------------------------------
// LayerNorm backward, the GEMM's column-sum epilogue and devias_colsum all end in the same second stage, out[i] = beta out[i] + sum_p part[p][i] in a
// fixed order, each as its own 6-12 us launch.  A fused region (csrc/regions.hip) that hands each of them a PRIVATE partial area collects those second
// stages in a list instead and runs them as ONE launch before it returns (devias_flush_deferred: the same arithmetic in the same order, bitwise).
struct DeviasReduceJob { const float* part; int nparts; int stride; int n; float* out; float beta; };
struct DeviasDeferList { enum { MAX = 16 }; DeviasReduceJob jobs[MAX]; int n; };
DeviasDeferList*& devias_defer_slot();                            // api.hip: thread-local; non-null only while a region is collecting
int& devias_defer_enabled();                                      // api.hip: option "regions_defer" / DEVIAS_REGIONS_DEFER (1, default; 0 = every second stage its own launch)
int devias_flush_deferred(DeviasDeferList* l, hipStream_t st);    // elementwise.hip
int64_t devias_layernorm_bwd_parts_count(int M);                  // layernorm.hip: partial rows that devias_layernorm_bwd_parts writes for M rows ([count][3][D] floats)
int devias_layernorm_bwd_parts(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, float* part,
                               int M, int D, int dtype, hipStream_t st);   // layernorm.hip: few rows, partials only (the caller reduces them, stacked over layers)
int devias_colsum_finish(const float* part, int nparts, int N, float* out, float beta, hipStream_t st);   // elementwise.hip: second stage of a column sum (deferred when a region collects)
int devias_row_scale_colsum(const void* x, const float* scale, int rps, void* y, int dtype, int M, int N, float* out, float beta, float* ws, hipStream_t st);   // elementwise.hip: y = x * scale[row / rps] and its column sums in one pass
// true = the `count` second stages described by `j` were taken over by the collecting region (the caller must NOT launch them)
static inline bool devias_defer(const DeviasReduceJob* j, int count) {
    DeviasDeferList* l = devias_defer_slot();
    if (!l || !devias_defer_enabled() || l->n + count > DeviasDeferList::MAX) return false;
    for (int i = 0; i < count; ++i) l->jobs[l->n++] = j[i];
    return true;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- scalar conversions -------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// ---- 4-element vector load/store of T as fp32 (address must be 4*sizeof(T)-aligned) --------------
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}

// ---- wave reductions (all 64 lanes participate) ---------------------------------------------------
// the 64-lane sum without the LDS crossbar: four DPP adds inside each row of 16 lanes (quad_perm, quad_perm, row_ror:4, row_ror:8), then the four row sums by v_permlane32_swap and
// v_permlane16_swap; every lane ends with the total.  Six `__shfl_xor` (ds_bpermute round trips) per sum were what held the LayerNorm kernels below the copy rate: forward 31.2 -> 26.6 us
// at [50176, 768] (5.8 TB/s), backward 61.2 -> 58.5 (profiles/r4_packed_fp32.txt).  The association differs from the xor butterfly's: sums change in their last bits, deterministically.
__device__ __forceinline__ float wave_sum(float v) {
#define DEVIAS_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    DEVIAS_DPP_ADD(0xB1); DEVIAS_DPP_ADD(0x4E); DEVIAS_DPP_ADD(0x124); DEVIAS_DPP_ADD(0x128);
#undef DEVIAS_DPP_ADD
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#define DEVIAS_DPP_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false)))
    DEVIAS_DPP_MAX(0xB1); DEVIAS_DPP_MAX(0x4E); DEVIAS_DPP_MAX(0x124); DEVIAS_DPP_MAX(0x128);
#undef DEVIAS_DPP_MAX
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// bf16 kernels: GELU and its derivative from two odd degree-17 polynomials in t = clamp(x, -4, 4) / 4 (least-squares fits at
// Chebyshev nodes, evaluated in fp32 Horner form; tools/fit_gelu_poly.py):
//   erf(x/sqrt2)                              |error| <= 6e-5  (scaled by 1 + 6e-5 and clamped to [-1, 1]: the tails are exact)
//   g(x) = erf(x/sqrt2)/2 + x*pdf(x)          |error| <= 5.3e-4 (dGELU(x) = 1/2 + g(x))
// both an order of magnitude below bf16 resolution (3.9e-3).  All-FMA, no transcendental: 9 packed-fp32 FMAs per PAIR of
// elements (v_pk_fma_f32) instead of ~14 VALU + rcp + exp2 per element (Abramowitz & Stegun 7.1.26, the previous version):
// the GELU epilogues are VALU-issue bound, so this is what sets their cost.  fp32 (parity) kernels use erff / expf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DEVIAS_ERF_POLY(u)   (((((((((2.681678368e+00f * u - 1.466233920e+01f) * u + 3.579562272e+01f) * u - 5.219407220e+01f) * u + \
    5.153296562e+01f) * u - 3.705950413e+01f) * u + 2.021369346e+01f) * u - 8.499460359e+00f) * u + 3.191358921e+00f))
#define DEVIAS_DGELU_POLY(u) (((((((((1.612753209e+01f * u - 8.526842075e+01f) * u + 1.980196598e+02f) * u - 2.676213605e+02f) * u + \
    2.352999263e+02f) * u - 1.420446591e+02f) * u + 5.973788243e+01f) * u - 1.694027150e+01f) * u + 3.190259248e+00f))
__device__ __forceinline__ float gelu_fast(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f) * 0.25f, u = t * t;
    const float e = __builtin_amdgcn_fmed3f(DEVIAS_ERF_POLY(u) * t * 1.00006f, -1.0f, 1.0f);
    return x * (0.5f + 0.5f * e);
}
__device__ __forceinline__ float dgelu_fast(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f) * 0.25f, u = t * t;
    return 0.5f + DEVIAS_DGELU_POLY(u) * t;
}
// the same arithmetic on two elements per instruction (bitwise equal to the scalar forms: same FMAs in the same order)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    t *= 0.25f;
    const f32x2 u = t * t;
    f32x2 e = DEVIAS_ERF_POLY(u) * t * 1.00006f;
    e = f32x2{__builtin_amdgcn_fmed3f(e[0], -1.0f, 1.0f), __builtin_amdgcn_fmed3f(e[1], -1.0f, 1.0f)};
    return x * (0.5f + 0.5f * e);
}
__device__ __forceinline__ f32x2 dgelu_fast2(f32x2 x) {
    f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    t *= 0.25f;
    const f32x2 u = t * t;
    return 0.5f + DEVIAS_DGELU_POLY(u) * t;
}

// exact erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

template <typename T> __device__ __forceinline__ float gelu_t(float x);
template <> __device__ __forceinline__ float gelu_t<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16>(float x) { return gelu_fast(x); }
template <typename T> __device__ __forceinline__ float dgelu_t(float x);
template <> __device__ __forceinline__ float dgelu_t<float>(float x) { return dgelu_f(x); }
template <> __device__ __forceinline__ float dgelu_t<bf16>(float x) { return dgelu_fast(x); }
