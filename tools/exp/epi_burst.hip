// experiment (not part of the product): is the GEMM epilogue's store burst bound per CU or by the chip?
// 256 workgroups x 512 threads alternate a "K loop" (spin for `busy` ticks of the 100 MHz clock) with the epilogue's store pattern
// (one 256 x 256 bf16 tile = 128 KiB per workgroup: 16 rows x 64 B per wave-instruction into an [M, 2304] bf16 matrix, `nout` outputs),
// (mode 0) all in phase, as the persistent GEMM's rounds are; (mode 1) with the workgroups' phases spread over the period;
// (active < 256) with only some workgroups alive.  Prints the mean time from the first store to (a) the last store ISSUED, (b) all stores complete.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__global__ __launch_bounds__(512) void k(char* __restrict__ out, int rounds, int busy, int mode, int active, int nout, long long out_stride,
                                          long long* __restrict__ stamps, int shape) {
    if ((int)blockIdx.x >= 256) return;
    // active < 256: keep workgroups spread over the XCDs (blockIdx % 8 = XCD)
    if (active < 256 && (int)(blockIdx.x >> 3) >= active / 8) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t0 = wall_clock64();
    const int period = busy + 450;
    const int off = mode == 1 ? (int)(((blockIdx.x * 97u) & 255u) * (unsigned)period / 256u) : 0;
    while (wall_clock64() - t0 < off) __builtin_amdgcn_s_sleep(1);
    long long issue = 0, done = 0;
    u4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int r = 0; r < rounds; ++r) {
        const long long a0 = wall_clock64();
        while (wall_clock64() - a0 < busy) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        const long long a = wall_clock64();
        const int t = r * 256 + blockIdx.x;
        for (int o = 0; o < nout; ++o) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // shape 0: the GEMM epilogue's 16 rows x 64 B per wave-instruction (two instructions per 128-byte line); shape 1: 8 rows x 128 B (whole lines);
                // shape 2: 4 rows x 256 B
                const int r16 = shape == 0 ? (wave >> 2) * 128 + (i >> 1) * 16 + (lane & 15) : shape == 1 ? (wave >> 2) * 128 + i * 8 + (lane >> 3)
                                                                                                         : (wave >> 2) * 128 + (i >> 1) * 8 + (i & 1) * 4 + (lane >> 4) + 0 * 0;
                char* dst = shape == 0 ? out + o * out_stride + ((size_t)(t / 9) * 256 + r16) * 4608 + (t % 9) * 512 + (wave & 3) * 128 + (i & 1) * 64 + (lane >> 4) * 16
                          : shape == 1 ? out + o * out_stride + ((size_t)(t / 9) * 256 + r16) * 4608 + (t % 9) * 512 + (wave & 3) * 128 + (lane & 7) * 16
                                       : out + o * out_stride + ((size_t)(t / 9) * 256 + (wave >> 1) * 64 + i * 4 + (lane >> 4)) * 4608 + (t % 9) * 512 + (wave & 1) * 256 + (lane & 15) * 16;
                *reinterpret_cast<u4*>(dst) = v;
                v.x += 1;
            }
        }
        const long long b = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long c = wall_clock64();
        issue += b - a;
        done += c - a;
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = issue;
        stamps[blockIdx.x * 4 + 1] = done;
        stamps[blockIdx.x * 4 + 2] = t0;
        stamps[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0;
    printf("store shape %d\n", shape);
    const int rounds = 9;
    const long long out_stride = (long long)(rounds * 256 / 9 + 1) * 256 * 4608;
    char* out; long long* st;
    (void)hipMalloc(&out, 2 * out_stride);
    (void)hipMalloc(&st, 256 * 4 * sizeof(long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-46s %10s %10s %12s %12s\n", "case", "issue us", "done us", "kernel us", "launch us");
    for (int nout = 1; nout <= 2; ++nout)
        for (int busy : {1800, 0})
            for (int cfg = 0; cfg < 5; ++cfg) {
                const int mode = cfg == 1 ? 1 : 0;
                const int active = cfg == 2 ? 128 : cfg == 3 ? 32 : cfg == 4 ? 8 : 256;
                if (busy == 0 && mode == 1) continue;
                float ms = 0;
                std::vector<long long> h(1024);
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipMemset(st, 0, 256 * 4 * sizeof(long long));
                    (void)hipEventRecord(e0);
                    for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, rounds, busy, mode, active, nout, out_stride, st, shape);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                (void)hipMemcpy(h.data(), st, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
                double is = 0, dn = 0; long long tmin = 1LL << 62, tmax = 0; int n = 0;
                for (int b = 0; b < 256; ++b) if (h[b * 4 + 3]) {
                    is += h[b * 4] * 0.01 / rounds; dn += h[b * 4 + 1] * 0.01 / rounds; ++n;
                    tmin = std::min(tmin, h[b * 4 + 2]); tmax = std::max(tmax, h[b * 4 + 3]);
                }
                char name[128];
                snprintf(name, sizeof name, "%d output(s), K loop %4.1f us, %s, %3d workgroups", nout, busy * 0.01, mode ? "phases spread" : "in phase     ", active);
                printf("%-46s %10.2f %10.2f %12.1f %12.1f\n", name, is / n, dn / n, (tmax - tmin) * 0.01, ms * 1e3 / 4);
            }
    return 0;
}
