// experiment: does an operand stream split between LDS-DMA (A tile) and register loads + ds_write_b128 (B tile) beat either alone?
// 256x256x64 tile pattern (A 32 KiB + B 32 KiB per K-tile), 2 LDS stages, one barrier per K-tile, XCD-chunked tile order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_p;
typedef const __attribute__((address_space(1))) void* glb_p;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
// MODE 0: A,B by DMA.  1: A by DMA, B through registers.  2: A,B through registers.
template <int MODE, int NT, int PF = 0, int TILED = 0>
__global__ __launch_bounds__(NT) void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int ld, int nk, int tiles_n, int ntiles, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
    constexpr int NW = NT / 64, PER = 32 / NW;            // 1-KiB DMA instructions per wave per operand
    constexpr int RPT = 2048 / NT;                         // 16-byte register pieces per thread per operand
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc = 0.f;
    auto dma = [&](const uint16_t* P, int r0, int ks, char* dst, bool tiled = false) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r8 = (wave * PER + i) * 8;
            const int row = r8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
            const uint16_t* src = tiled ? P + ((int64_t)(r0 >> 8) * nk + ks) * 16384 + row * 64 + chunk * 8 : P + (int64_t)(r0 + row) * ld + ks * 64 + chunk * 8;
            __builtin_amdgcn_global_load_lds((glb_p)src, (lds_p)(dst + r8 * 128), 16, 0, 0);
        }
    };
    auto rload = [&](const uint16_t* P, int r0, int ks, u4 (&v)[RPT]) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = (tid >> 3) + i * (NT / 8), chunk = tid & 7;
            v[i] = *reinterpret_cast<const u4*>(P + (int64_t)(r0 + row) * ld + ks * 64 + chunk * 8);
        }
    };
    auto rstore = [&](char* dst, const u4 (&v)[RPT]) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = (tid >> 3) + i * (NT / 8), chunk = (tid & 7) ^ (row & 7);
            *reinterpret_cast<u4*>(dst + row * 128 + chunk * 16) = v[i];
        }
    };
    for (int t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
        int q = ntiles >> 3, r = ntiles & 7, x = t0 & 7;
        int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t0 >> 3);
        const int tm = t / tiles_n, tn = t % tiles_n;
        u4 va[RPT], vb[RPT]; unsigned pf = 0;
        if (PF > 0 && wave < 4) { pf = *reinterpret_cast<const unsigned*>(A + (int64_t)(tm * 256 + wave * 64 + lane) * ld + 64); asm volatile("" :: "v"(pf)); }
        if (MODE <= 1) dma(A, tm * 256, 0, smem, TILED); else rload(A, tm * 256, 0, va);
        if (MODE == 0) dma(B, tn * 256, 0, smem + 32768); else rload(B, tn * 256, 0, vb);
        for (int ks = 0; ks < nk; ++ks) {
            char* cur = smem + (ks & 1) * 65536;
            char* nxt = smem + ((ks + 1) & 1) * 65536;
            if (PF > 0 && wave < 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2) rstore(cur, va);
            if (MODE >= 1) rstore(cur + 32768, vb);
            __syncthreads();
            if (ks + 1 < nk) {
                if (MODE <= 1) dma(A, tm * 256, ks + 1, nxt, TILED); else rload(A, tm * 256, ks + 1, va);
                if (MODE == 0) dma(B, tn * 256, ks + 1, nxt + 32768); else rload(B, tn * 256, ks + 1, vb);
            }
            if (PF > 0 && wave < 4) {
                int kp = ks + PF < nk ? ks + PF : nk - 1;
                pf = *reinterpret_cast<const unsigned*>(A + (int64_t)(tm * 256 + wave * 64 + lane) * ld + kp * 64);
                asm volatile("" :: "v"(pf));
            }
            acc += reinterpret_cast<float*>(cur)[tid] + reinterpret_cast<float*>(cur + 32768)[tid];
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE, int NT, int PF = 0, int TILED = 0> void run(const uint16_t* A, const uint16_t* B, int M, int N, int K, float* out, int ld = 0) {
    if (!ld) ld = K;
    const int nk = K / 64, tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NT, PF, TILED>), dim3(256), dim3(NT), 0, 0, A, B, ld, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<MODE, NT, PF, TILED>), dim3(256), dim3(NT), 0, 0, A, B, ld, nk, tiles_n, ntiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double bytes = (double)ntiles * nk * 65536.0;
    printf("tiled=%d M=%d pf=%d ld=%d mode=%d threads=%d N=%d K=%d: %7.1f us  %6.2f TB/s (%5.1f GB/s per CU) == %5.0f TFLOP/s\n", TILED, M, PF, ld, MODE, NT, N, K, ms * 1e3,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    uint16_t *A, *B; float* out;
    (void)hipMalloc(&A, (size_t)50176 * 4096 * 2); (void)hipMalloc(&B, (size_t)4096 * 4096 * 2); (void)hipMalloc(&out, 64);
    (void)hipMemset(A, 0, (size_t)50176 * 4096 * 2); (void)hipMemset(B, 0, (size_t)4096 * 4096 * 2);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 512, 0, 0>(A, B, 50176, 768, 3072, out); run<0, 512, 0, 1>(A, B, 50176, 768, 3072, out);
        run<0, 512, 0, 0>(A, B, 12544, 768, 3072, out); run<0, 512, 0, 1>(A, B, 12544, 768, 3072, out);
        run<0, 512, 0, 0>(A, B, 50176, 768, 1536, out); run<0, 512, 0, 1>(A, B, 50176, 768, 1536, out);
        run<0, 512, 0, 0>(A, B, 50176, 768, 768, out); run<0, 512, 0, 1>(A, B, 50176, 768, 768, out);
        run<0, 512, 0, 0>(A, B, 50176, 2304, 768, out); run<0, 512, 0, 1>(A, B, 50176, 2304, 768, out);
        run<0, 512, 0, 0>(A, B, 50176, 3072, 3072, out); run<0, 512, 0, 1>(A, B, 50176, 3072, 3072, out);
    }
    return 0;
}
