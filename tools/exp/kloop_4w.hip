// experiment: the vendor library's structure for MT256x256x64 -- 4 waves per workgroup (one per SIMD), 128x128 per wave, 256 accumulator registers (AGPRs) --
// against this library's 8 waves of 128x64: K-tile time with the fragment reads (no LDS-DMA, no epilogue).
//   per wave and 64-deep K-tile: 2 k-steps x (8 A + 8 B fragments = 16 ds_read_b128, 64 MFMA 16x16x32); reads of k-step s + 1 issued during the MFMAs of k-step s
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int SPREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 fa[8], fb[8], ga[8], gb[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { fa[i][e] = (__bf16)(0.01f * (i + e)); fb[i][e] = (__bf16)(0.02f * (i - e)); ga[i] = fa[i]; gb[i] = fb[i]; }
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const char* st = smem + (it & 1) * 65536 + wave * 8192;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 (&ca)[8] = ks == 0 ? fa : ga; bf16x8 (&cb)[8] = ks == 0 ? fb : gb;
            bf16x8 (&na)[8] = ks == 0 ? ga : fa; bf16x8 (&nb)[8] = ks == 0 ? gb : fb;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[q], acc[q][j], 0, 0, 0);
                if (SPREAD) {
                    na[q] = *reinterpret_cast<const bf16x8*>(st + ((ks * 16 + q) * 1024) % 32768 + lane * 16);
                    nb[q] = *reinterpret_cast<const bf16x8*>(st + 32768 + ((ks * 16 + 8 + q) * 1024) % 24576 + lane * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}
template <int SPREAD> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SPREAD>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SPREAD>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 2000;
    float t0 = run<0>(out, iters), t1 = run<1>(out, iters);
    printf("4 waves x 128x128, MFMA only              %.3f us per 256x256x64 K-tile (%.0f TFLOP/s chip-wide)\n", t0 * 1e3 / iters, 256.0 * 2 * 256 * 256 * 64 / (t0 * 1e-3 / iters) / 1e12);
    printf("4 waves x 128x128, reads a k-step ahead   %.3f us per 256x256x64 K-tile (%.0f TFLOP/s chip-wide)\n", t1 * 1e3 / iters, 256.0 * 2 * 256 * 256 * 64 / (t1 * 1e-3 / iters) / 1e12);
    return 0;
}
