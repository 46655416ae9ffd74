// experiment: do MFMA and VALU (+ transcendental) work overlap on one SIMD -- across waves, and within one wave when interleaved?
// per iteration and wave: 32 x v_mfma_f32_16x16x32_bf16 (independent accumulators) and/or a softmax-like VALU block (32 v_exp_f32 + 96 v_fma_f32)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>   // 0 mfma only, 1 valu only, 2 both in blocks, 3 both interleaved (1 mfma : 4 valu)
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = seed * (i + 1) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float x = v[i];
                x = __builtin_fmaf(x, 0.999f, -0.001f);
                x = __builtin_amdgcn_exp2f(x);
                x = __builtin_fmaf(x, 0.5f, 0.25f);
                x = __builtin_fmaf(x, 0.75f, -0.125f);
                v[i] = x;
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> void run(float* out, int blocks_per_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves, each `iters` iterations
    printf("mode=%d waves/SIMD=%d: %8.1f us  -> %7.1f ns per iteration-wave per SIMD (%.0f clk @2.4GHz)\n", MODE, blocks_per_cu, ms * 1e3,
           ms * 1e6 / iters / blocks_per_cu, ms * 1e6 / iters / blocks_per_cu * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 64);
    for (int w = 1; w <= 3; ++w) { run<0>(out, w); run<1>(out, w); run<2>(out, w); run<3>(out, w); }
    return 0;
}
