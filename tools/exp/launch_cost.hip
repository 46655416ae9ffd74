// experiment: per-block fixed cost vs LDS size / threads per block / first-touch loads (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int LDS, int NT, int MODE>
__global__ __launch_bounds__(NT) void k(const float* __restrict__ in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    float acc = 0.f;
    if (MODE >= 1) {   // each thread reads 128 B from a block-contiguous 64 KB region, via registers -> LDS
        const float4* p = reinterpret_cast<const float4*>(in) + (size_t)blockIdx.x * 4096 + threadIdx.x;
        float4 v0 = p[0], v1 = p[NT], v2 = p[2 * NT], v3 = p[3 * NT];
        float4* s = reinterpret_cast<float4*>(smem);
        s[threadIdx.x] = v0; s[threadIdx.x + NT] = v1;
        acc = v2.x + v3.y;
    }
    __syncthreads();
    for (int i = 0; i < iters; ++i) acc += reinterpret_cast<float*>(smem)[(threadIdx.x + i * 64) & (LDS / 4 - 1)];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
template <int LDS, int NT, int MODE>
void run(const char* name, const float* in, float* out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<LDS, NT, MODE>), dim3(blocks), dim3(NT), 0, 0, in, out, 8);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<LDS, NT, MODE>), dim3(blocks), dim3(NT), 0, 0, in, out, 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s blocks=%5d  %8.2f us/launch  %7.1f ns/block\n", name, blocks, ms / 20 * 1e3, ms / 20 * 1e6 / blocks);
}
int main() {
    float *in, *out; hipMalloc(&in, (size_t)8192 * 65536); hipMalloc(&out, 1 << 20);
    hipMemset(in, 0, (size_t)8192 * 65536);
    for (int blocks : {256, 1764, 7056}) {
        run<131072, 512, 0>("LDS128K T512 noload", in, out, blocks);
        run<65536, 512, 0>("LDS64K  T512 noload", in, out, blocks);
        run<65536, 256, 0>("LDS64K  T256 noload", in, out, blocks);
        run<32768, 256, 0>("LDS32K  T256 noload", in, out, blocks);
        run<1024, 256, 0>("LDS1K   T256 noload", in, out, blocks);
        run<131072, 512, 1>("LDS128K T512 load64K", in, out, blocks);
        run<65536, 256, 1>("LDS64K  T256 load64K", in, out, blocks);
        run<32768, 256, 1>("LDS32K  T256 load64K", in, out, blocks);
    }
    return 0;
}
