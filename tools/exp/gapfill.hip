// experiment: what hides in the gap behind a v_mfma_f32_32x32x16_bf16 issued by ONE wave per SIMD, by the form of the MFMA's operands?
//   FORM 0: D (AGPR) += A (VGPR) x B (VGPR)          -- the dK / dV / dQ accumulation MFMAs of the one-wave attention kernels
//   FORM 1: D (VGPR)  = A (VGPR) x B (AGPR) + D      -- their S / dP MFMAs
//   FORM 2: D (AGPR) += A (VGPR) x B (AGPR)
// DEP = how many other MFMAs sit between two MFMAs on the same accumulator (1: S / dP chains of the kernels; 3: their accumulation MFMAs)
// fillers per gap: NF instructions of kind FT (0 v_mul_f32, 1 v_exp_f32, 2 alternating exp / mul / cvt_pk as in the softmax arithmetic, 3 ds_read_b128, 4 ds_read_b64_tr_b16,
// 5 = NF softmax-mix instructions + one ds_read_b128 + one ds_read_b64_tr_b16), independent of the MFMAs; LDS reads are waited for once per 16 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int FORM, int DEP, int NF, int FT>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(smem)[i] = seed * i;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4; typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x4 lr[4]; u32x2 lt[4];
    for (int i = 0; i < 4; ++i) { lr[i] = u32x4{0, 0, 0, 0}; lt[i] = u32x2{0, 0}; }
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    asm volatile("" ::: "a0", "a15", "a16", "a31", "a32", "a47", "a48", "a63", "a64", "a67");
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    unsigned w0 = __float_as_uint(seed);
    asm volatile("v_accvgpr_write_b32 a64, %0\n v_accvgpr_write_b32 a65, %0\n v_accvgpr_write_b32 a66, %0\n v_accvgpr_write_b32 a67, %0" :: "v"(w0));
    for (int i = 0; i < 64; ++i) asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(0));
    f32x16 d[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) d[i][j] = 0.f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed * (i + 1) * 1e-3f;
    unsigned pk = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            constexpr int NACC = DEP + 1;
            const int acc = m % NACC;
            if (FORM == 0) { if (acc == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b)); else if (acc == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a), "v"(b)); else if (acc == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a), "v"(b)); }
            else if (FORM == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[64:67], %0" : "+v"(d[acc]) : "v"(a));
            else { if (acc == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, a[64:67], a[0:15]" :: "v"(a)); else if (acc == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, a[64:67], a[16:31]" :: "v"(a)); else if (acc == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, a[64:67], a[32:47]" :: "v"(a)); else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, a[64:67], a[48:63]" :: "v"(a)); }
            if (FT == 3) { _Pragma("unroll") for (int n = 0; n < NF; ++n) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(lr[(m * NF + n) & 3]) : "v"(laddr), "i"(((m * NF + n) & 3) * 1024)); }
            if (FT == 4) { _Pragma("unroll") for (int n = 0; n < NF; ++n) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(lt[(m * NF + n) & 3]) : "v"(laddr), "i"(((m * NF + n) & 3) * 1024)); }
            if (FT == 5) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(lr[m & 3]) : "v"(laddr), "i"((m & 3) * 1024)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(lt[m & 3]) : "v"(laddr), "i"((m & 3) * 1024 + 512)); }
#pragma unroll
            for (int n = 0; n < (FT == 3 || FT == 4 ? 0 : NF); ++n) {
                const int r = (m * NF + n) & 7;
                if (FT == 0) f[r] = f[r] * 0.999f;
                else if (FT == 1) f[r] = __builtin_amdgcn_exp2f(f[r]);
                else { const int q = (m * NF + n) % 3; if (q == 0) f[r] = __builtin_amdgcn_exp2f(f[r]); else if (q == 1) f[r] = f[r] * f[(r + 1) & 7]; else { typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2; const bf16x2 t = {(__bf16)f[r], (__bf16)f[(r + 3) & 7]}; pk ^= *reinterpret_cast<const unsigned*>(&t); } }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FT >= 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lr[0]), "+v"(lr[1]), "+v"(lr[2]), "+v"(lr[3]), "+v"(lt[0]), "+v"(lt[1]), "+v"(lt[2]), "+v"(lt[3]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][7] + lr[i][0] + lt[i][1];
    if (s == 12345.678f || pk == 0x12345u) out[4096] = pk;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int FORM, int DEP, int NF, int FT> void run(unsigned long long* out) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<FORM, DEP, NF, FT>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("form %d dep %d: %d fillers of kind %d per gap -> %6.1f cycles per MFMA gap (median wave; p10 %6.1f, p90 %6.1f)\n", FORM, DEP, NF, FT, h[512] / (double)(iters * 16),
           h[102] / (double)(iters * 16), h[921] / (double)(iters * 16));
}
int main() {
    unsigned long long* out; (void)hipMalloc(&out, 8192 * 8);
    run<0, 3, 1, 3>(out); run<0, 3, 2, 3>(out); run<0, 3, 1, 4>(out); run<0, 3, 2, 4>(out); run<0, 3, 3, 5>(out); run<1, 1, 1, 3>(out); run<1, 1, 2, 4>(out); run<1, 1, 3, 5>(out); run<1, 1, 0, 5>(out); run<0, 3, 0, 5>(out);
    run<0, 3, 0, 0>(out); run<0, 3, 2, 0>(out); run<0, 3, 4, 0>(out); run<0, 3, 6, 0>(out); run<0, 3, 3, 2>(out); run<0, 3, 5, 2>(out); run<0, 3, 2, 1>(out);
    run<1, 1, 0, 0>(out); run<1, 1, 2, 0>(out); run<1, 1, 4, 0>(out); run<1, 1, 6, 0>(out); run<1, 1, 3, 2>(out); run<1, 1, 5, 2>(out); run<1, 1, 2, 1>(out);
    run<1, 3, 0, 0>(out); run<1, 3, 3, 2>(out); run<1, 3, 5, 2>(out);
    run<2, 3, 0, 0>(out); run<2, 3, 3, 2>(out); run<2, 3, 5, 2>(out);
    return 0;
}
