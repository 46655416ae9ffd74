// experiment: LDS-DMA stream rate per CU vs bytes in flight (ring of NS stages of 32 KiB = a 256x32 A tile + 256x32 B tile; NS-1 stages in flight)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_p;
typedef const __attribute__((address_space(1))) void* glb_p;
template <int N> __device__ __forceinline__ void waitv() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
template <int NS, int NT, int LOC, int AUX>
__global__ __launch_bounds__(NT) void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int ld, int nk, int tiles_n, int ntiles, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[NS * 32768];
    constexpr int NW = NT / 64, PER = 16 / NW;            // 1-KiB instructions per wave per operand per stage (16 KiB per operand)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc = 0.f;
    auto issue = [&](int tm, int tn, int ks, int stage) {
        char* st = smem + stage * 32768;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r16 = (wave * PER + i) * 16;
            const int row = r16 + (lane >> 2), chunk = lane & 3;
            const uint16_t* sa = A + (int64_t)(tm * 256 + row) * ld + ks * 32 + chunk * 8;
            const uint16_t* sb = B + (int64_t)(tn * 256 + row) * ld + ks * 32 + chunk * 8;
            __builtin_amdgcn_global_load_lds((glb_p)sa, (lds_p)(st + r16 * 64), 16, 0, AUX);
            __builtin_amdgcn_global_load_lds((glb_p)sb, (lds_p)(st + 16384 + r16 * 64), 16, 0, AUX);
        }
    };
    for (int t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
        int t = t0;
        if (LOC == 1) { int q = ntiles >> 3, r = ntiles & 7, x = t0 & 7; t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t0 >> 3); }
        if (LOC == 2) t = (t0 & 7) + 8 * (t0 / 256);
        const int tm = t / tiles_n, tn = t % tiles_n;
        for (int s = 0; s < NS - 1 && s < nk; ++s) issue(tm, tn, s, s);
        int cs = 0;
        for (int ks = 0; ks < nk; ++ks) {
            // (NS-2) younger stages may stay in flight
            if (ks + NS - 2 < nk) waitv<(NS - 2) * 2 * PER>(); else waitv<0>();
            __builtin_amdgcn_s_barrier();
            if (ks + NS - 1 < nk) { int ns = cs + NS - 1; if (ns >= NS) ns -= NS; issue(tm, tn, ks + NS - 1, ns); }
            acc += reinterpret_cast<float*>(smem + cs * 32768)[threadIdx.x];
            if (++cs == NS) cs = 0;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int NS, int NT, int LOC, int AUX> void run(const uint16_t* A, const uint16_t* B, int M, int N, int K, float* out) {
    const int nk = K / 32, tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NS, NT, LOC, AUX>), dim3(256), dim3(NT), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<NS, NT, LOC, AUX>), dim3(256), dim3(NT), 0, 0, A, B, K, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double bytes = (double)ntiles * nk * 32768.0;
    printf("aux=%d loc=%d stages=%d (%3d KiB in flight) threads=%d N=%d K=%d: %7.1f us  %6.2f TB/s (%5.1f GB/s per CU) == %5.0f TFLOP/s\n", AUX, LOC, NS, (NS - 1) * 32, NT, N, K, ms * 1e3,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    uint16_t *A, *B; float* out;
    (void)hipMalloc(&A, (size_t)50176 * 3072 * 2); (void)hipMalloc(&B, (size_t)4096 * 3072 * 2); (void)hipMalloc(&out, 64);
    (void)hipMemset(A, 0, (size_t)50176 * 3072 * 2); (void)hipMemset(B, 0, (size_t)4096 * 3072 * 2);
    for (int rep = 0; rep < 2; ++rep) {
        run<3, 512, 1, 0>(A, B, 50176, 2304, 768, out);
        run<3, 512, 1, 1>(A, B, 50176, 2304, 768, out);
        run<3, 512, 1, 2>(A, B, 50176, 2304, 768, out);
        run<3, 512, 1, 16>(A, B, 50176, 2304, 768, out);
        run<3, 512, 1, 17>(A, B, 50176, 2304, 768, out);
    }
    return 0;
}
