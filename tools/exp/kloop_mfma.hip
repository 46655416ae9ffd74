// experiment: is the 256x256x64 K-loop bound by the SIMD's vector-issue port rather than by the MFMA pipe?
// 8 waves (2 per SIMD) per workgroup, one workgroup per CU, 128 KiB of LDS, one barrier per iteration; per wave and iteration the LDS reads and MFMAs of a
// 128x64x64 wave tile:   MODE 0: 64 x v_mfma_f32_16x16x32_bf16 (8 issue cycles of 16) + 24 ds_read_b128
//                        MODE 1: 32 x v_mfma_f32_32x32x16_bf16 (8 issue cycles of 32) + 24 ds_read_b128
//                        MODE 2 / 3: the same without the LDS reads (operands held in registers)
//                        MODE 4: as 0, but the 12 reads of k-step s + 1 are issued BEFORE the 32 MFMAs of k-step s (two register sets)
//                        MODE 5: as 4, reads spread: 3 reads after every 8 MFMAs
//                        MODE 6: as 5, with a workgroup barrier per k-step instead of per K-tile
// prints the time per iteration; the LDS addressing is conflict-free in every mode (each 16-lane group of a ds_read_b128 covers 256 contiguous bytes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 acc4[8][4];
    f32x16 acc16[4][2];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc16[i][j][e] = 0.f;
    bf16x8 fa[8], fb[4];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) fa[i][e] = (__bf16)(0.01f * (i + e));
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) fb[i][e] = (__bf16)(0.02f * (i - e));
    if (MODE >= 4) {
        bf16x8 ga[8], gb[4];
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            const char* st = smem + (it & 1) * 65536 + wave * 4096;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (MODE == 6 && ks == 1) __builtin_amdgcn_s_barrier();     // MODE 6: a workgroup barrier per k-step (BK = 32 ring)
                // issue the reads of the next k-step into the other register set, then multiply the current one
                bf16x8 (&ca)[8] = ks == 0 ? fa : ga; bf16x8 (&cb)[4] = ks == 0 ? fb : gb;
                bf16x8 (&na)[8] = ks == 0 ? ga : fa; bf16x8 (&nb)[4] = ks == 0 ? gb : fb;
                if (MODE == 4) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) na[i] = *reinterpret_cast<const bf16x8*>(st + (ks * 12 + i) * 1024 % 32768 + lane * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(st + 32768 + ((ks * 12 + 8 + j) * 1024) % 28672 + lane * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc4[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int i = 2 * q; i < 2 * q + 2; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc4[i][j], 0, 0, 0);
                        na[2 * q] = *reinterpret_cast<const bf16x8*>(st + (ks * 12 + 2 * q) * 1024 % 32768 + lane * 16);
                        na[2 * q + 1] = *reinterpret_cast<const bf16x8*>(st + (ks * 12 + 2 * q + 1) * 1024 % 32768 + lane * 16);
                        nb[q] = *reinterpret_cast<const bf16x8*>(st + 32768 + ((ks * 12 + 8 + q) * 1024) % 28672 + lane * 16);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const char* st = smem + (it & 1) * 65536 + wave * 4096;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE < 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + (ks * 12 + i) * 1024 % 32768 + lane * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + 32768 + ((ks * 12 + 8 + j) * 1024) % 28672 + lane * 16);
            }
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc4[i][j], 0, 0, 0);
            } else {
                // the same operand registers feed two 16-deep k-steps of 32x32x16 (4 x 2 blocks of 32 x 32 per step)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2 * h + j], fa[4 * h + i], acc16[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc4[i][j][0] + acc4[i][j][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc16[i][j][0] + acc16[i][j][15];
    if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}
template <int MODE> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    const char* names[7] = {"16x16x32 + LDS reads", "32x32x16 + LDS reads", "16x16x32 only", "32x32x16 only", "16x16x32, reads 1 step ahead", "16x16x32, reads ahead + spread", "... + barrier per k-step"};
    float t[7] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters), run<6>(out, iters)};
    for (int m = 0; m < 7; ++m) printf("%-32s %.3f us per 256x256x64 K-tile   (%.0f TFLOP/s chip-wide)\n", names[m], t[m] * 1e3 / iters, 256.0 * 2 * 256 * 256 * 64 / (t[m] * 1e-3 / iters) / 1e12);
    return 0;
}
