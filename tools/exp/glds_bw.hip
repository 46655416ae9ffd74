// experiment (not product): per-CU operand-stream ceiling of the GEMM loader.  256 workgroups x 512 threads, each streams 256-row x 64-col
// bf16 tiles (32 KiB) for A and B into a 2-stage LDS ring with the gemm256 loop structure (wait vmcnt(0), barrier, issue next), no MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_p;
typedef const __attribute__((address_space(1))) void* glb_p;

template <int MODE>   // 0: row-strided (ld elements per row), 1: tile-contiguous (each 1 KiB instruction reads 1 KiB contiguous),
                      // 2: row-strided in HALF lines (16 rows x 64 B per instruction: the 32-deep units of a four-slot ring)
__global__ __launch_bounds__(512) void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int ld, int nk, int tiles_n, int ntiles, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc = 0.f;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        for (int kt = 0; kt <= nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt < nk) {
                char* st = smem + (kt & 1) * 65536;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r8 = wave * 32 + i * 8;
                    const uint16_t *sa, *sb;
                    if (MODE == 2) {
                        // instruction i of 4: k-half i >> 1, rows (wave * 2 + (i & 1)) * 16 + (lane >> 2), 16-byte chunk lane & 3 of that half
                        const int row = (wave * 2 + (i & 1)) * 16 + (lane >> 2), half = i >> 1;
                        sa = A + (int64_t)(tm * 256 + row) * ld + kt * 64 + half * 32 + (lane & 3) * 8;
                        sb = B + (int64_t)(tn * 256 + row) * ld + kt * 64 + half * 32 + (lane & 3) * 8;
                    } else if (MODE == 0) {
                        const int row = r8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
                        sa = A + (int64_t)(tm * 256 + row) * ld + kt * 64 + chunk * 8;
                        sb = B + (int64_t)(tn * 256 + row) * ld + kt * 64 + chunk * 8;
                    } else {
                        sa = A + ((int64_t)(tm * nk + kt) * 32 + wave * 4 + i) * 512 + lane * 8;
                        sb = B + ((int64_t)(tn * nk + kt) * 32 + wave * 4 + i) * 512 + lane * 8;
                    }
                    __builtin_amdgcn_global_load_lds((glb_p)sa, (lds_p)(st + r8 * 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((glb_p)sb, (lds_p)(st + 32768 + r8 * 128), 16, 0, 0);
                }
            }
            if (kt > 0) acc += reinterpret_cast<float*>(smem + ((kt - 1) & 1) * 65536)[threadIdx.x];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE> void run(const char* name, const uint16_t* A, const uint16_t* B, int M, int N, int K, float* out) {
    const int nk = K / 64, tiles_m = M / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double bytes = (double)ntiles * nk * 65536.0;
    printf("%-28s M=%d N=%d K=%d: %7.1f us  L2->LDS %6.2f TB/s  (%5.1f GB/s per CU)  == %6.0f TFLOP/s if MFMA kept up\n", name, M, N, K, ms * 1e3,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    uint16_t *A, *B; float* out;
    (void)hipMalloc(&A, (size_t)50176 * 3072 * 2); (void)hipMalloc(&B, (size_t)4096 * 3072 * 2); (void)hipMalloc(&out, 64);
    (void)hipMemset(A, 0, (size_t)50176 * 3072 * 2); (void)hipMemset(B, 0, (size_t)4096 * 3072 * 2);
    run<0>("row-strided qkv", A, B, 50176, 2304, 768, out);
    run<2>("row-strided half lines qkv", A, B, 50176, 2304, 768, out);
    run<1>("tile-contiguous qkv", A, B, 50176, 2304, 768, out);
    run<0>("row-strided fc1", A, B, 50176, 3072, 768, out);
    run<1>("tile-contiguous fc1", A, B, 50176, 3072, 768, out);
    run<0>("row-strided fc2 (K=3072)", A, B, 50176, 768, 3072, out);
    run<1>("tile-contiguous fc2", A, B, 50176, 768, 3072, out);
    return 0;
}
