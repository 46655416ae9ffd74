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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// bf16 kernels: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, three orders of magnitude below bf16 resolution)
// sharing ONE exponential, E = exp(-x^2/2), between erf(x/sqrt2) and the Gaussian density of the derivative:
//   ~20 VALU + 2 transcendental ops per element instead of the ~60 of erff()+expf().  fp32 (parity) kernels use erff.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float E = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);          // exp(-x^2/2)
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float erfz = 1.0f - poly * t * E;                                             // erf(|x|/sqrt2)
    cdf = 0.5f * (1.0f + copysignf(erfz, x));
    pdf = 0.39894228040143267794f * E;
}
__device__ __forceinline__ float gelu_fast(float x) { float c, d; gelu_parts_fast(x, c, d); return x * c; }
__device__ __forceinline__ float dgelu_fast(float x) { float c, d; gelu_parts_fast(x, c, d); return c + x * d; }

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
