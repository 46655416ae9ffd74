// experiment: operand stream through REGISTERS (global_load_dwordx4 -> ds_write_b128) vs LDS-DMA, same tile pattern, XCD-chunked order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
template <int NT, int DEPTH, bool WRITE_LDS>
__global__ __launch_bounds__(NT) void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int ld, int nk, int tiles_n, int ntiles, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    constexpr int PER = 65536 / 16 / NT;                 // 16-byte chunks per thread per 64-deep K-tile (A 32 KiB + B 32 KiB)
    const int tid = threadIdx.x;
    u4 acc = {0, 0, 0, 0};
    for (int t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
        int q = ntiles >> 3, r = ntiles & 7, x = t0 & 7;
        int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t0 >> 3);
        const int tm = t / tiles_n, tn = t % tiles_n;
        u4 v[DEPTH][PER];
        auto load = [&](int kt, u4 (&d)[PER]) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int c = tid + i * NT;                    // chunk id in [0, 4096): first 2048 = A, then B
                int isb = c >> 11, cc = c & 2047, row = cc >> 3, ch = cc & 7;
                const uint16_t* base = isb ? B + (int64_t)(tn * 256 + row) * ld : A + (int64_t)(tm * 256 + row) * ld;
                d[i] = *reinterpret_cast<const u4*>(base + kt * 64 + ch * 8);
            }
        };
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) if (s < nk) load(s, v[s]);
        for (int kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                if (kt + s < nk) {
                    if (WRITE_LDS) {
                        __syncthreads();
#pragma unroll
                        for (int i = 0; i < PER; ++i) *reinterpret_cast<u4*>(smem + (tid + i * NT) * 16) = v[s][i];
                        __syncthreads();
                        acc.x += reinterpret_cast<unsigned*>(smem)[tid];
                    } else {
#pragma unroll
                        for (int i = 0; i < PER; ++i) acc += v[s][i];
                    }
                    if (kt + s + DEPTH < nk) load(kt + s + DEPTH, v[s]);
                }
            }
        }
    }
    if (acc.x == 12345u && acc.y == 777u) out[0] = 1.f;
}
template <int NT, int DEPTH, bool W> void run(const uint16_t* A, const uint16_t* B, int M, int N, int K, float* out, int blocks) {
    const int nk = K / 64, tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NT, DEPTH, W>), dim3(blocks), dim3(NT), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<NT, DEPTH, W>), dim3(blocks), dim3(NT), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double bytes = (double)ntiles * nk * 65536.0;
    printf("threads=%d blocks=%d depth=%d lds=%d N=%d K=%d: %7.1f us  %6.2f TB/s (%5.1f GB/s per CU) == %5.0f TFLOP/s\n", NT, blocks, DEPTH, (int)W, N, K, ms * 1e3,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    uint16_t *A, *B; float* out;
    (void)hipMalloc(&A, (size_t)50176 * 3072 * 2); (void)hipMalloc(&B, (size_t)4096 * 3072 * 2); (void)hipMalloc(&out, 64);
    (void)hipMemset(A, 0, (size_t)50176 * 3072 * 2); (void)hipMemset(B, 0, (size_t)4096 * 3072 * 2);
    for (int rep = 0; rep < 2; ++rep) {
        run<512, 1, false>(A, B, 50176, 2304, 768, out, 256);
        run<512, 2, false>(A, B, 50176, 2304, 768, out, 256);
        run<512, 2, true>(A, B, 50176, 2304, 768, out, 256);
        run<512, 2, false>(A, B, 50176, 2304, 768, out, 512);
        run<256, 2, false>(A, B, 50176, 2304, 768, out, 1024);
        run<512, 2, false>(A, B, 50176, 768, 3072, out, 256);
    }
    return 0;
}
