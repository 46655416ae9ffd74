// experiment: per-CU store rate of the GEMM epilogue pattern (8 rows x 128 B per wave-instruction at a 4608-B row stride) vs contiguous,
// with and without non-temporal / cache-policy hints; 256 workgroups x 512 threads, each wave stores 16 KiB per "tile"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
template <int PAT, int NT>
__global__ __launch_bounds__(512) void k(char* __restrict__ out, int tiles, int ld_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int t = 0; t < tiles; ++t) {
        // a 256 x 256 bf16 tile (128 KiB) per workgroup; wave w owns rows [128*(w>>2), +128) x 128-byte column block (w&3)
        char* tile = out + ((size_t)blockIdx.x * tiles + t) * 131072;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            char* dst;
            if (PAT == 0) dst = tile + wave * 16384 + i * 1024 + lane * 16;                               // 1 KiB contiguous per instruction
            else {
                const int row = (wave >> 2) * 128 + i * 8 + (lane >> 3);
                dst = tile + (size_t)row * 512 + (wave & 3) * 128 + (lane & 7) * 16;                          // 8 rows x 128 B, 512-B row stride inside the tile window
                if (PAT == 3) {   // 16 rows x 64 B per instruction (two 32-B runs from lane pairs), [M,2304] layout
                    const int r16 = (wave >> 2) * 128 + (i >> 1) * 16 + (lane & 15);
                    dst = out + ((size_t)((blockIdx.x * tiles + t) / 9) * 256 + r16) * (size_t)ld_bytes + ((blockIdx.x * tiles + t) % 9) * 512 + (wave & 3) * 128 + (i & 1) * 64 + (lane >> 4) * 16;
                }
                if (PAT == 2) dst = out + ((size_t)((blockIdx.x * tiles + t) / 9) * 256 + row) * (size_t)ld_bytes + ((blockIdx.x * tiles + t) % 9) * 512 + (wave & 3) * 128 + (lane & 7) * 16;  // real [M, 2304] layout
            }
            if (NT == 1) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst));
            else *reinterpret_cast<u4*>(dst) = v;
            v.x += 1;
        }
    }
}
template <int PAT, int NT> void run(char* out, const char* name) {
    const int tiles = 7;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PAT, NT>), dim3(256), dim3(512), 0, 0, out, tiles, 4608);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PAT, NT>), dim3(256), dim3(512), 0, 0, out, tiles, 4608);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double bytes = 256.0 * tiles * 131072;
    printf("%-44s nt=%d: %7.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", name, NT, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
    char* out; (void)hipMalloc(&out, (size_t)512 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(out, "1 KiB contiguous per instruction"); run<0, 1>(out, "1 KiB contiguous per instruction");
        run<1, 0>(out, "8 rows x 128 B, tile-local rows"); run<1, 1>(out, "8 rows x 128 B, tile-local rows");
        run<2, 0>(out, "8 rows x 128 B, [M,2304] bf16 rows"); run<3, 0>(out, "16 rows x 64 B, [M,2304] bf16 rows");
    }
    return 0;
}
