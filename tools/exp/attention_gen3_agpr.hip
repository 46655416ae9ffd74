// bf16 attention backward dK / dV for gfx950, one wave per SIMD with the whole register file: 64 keys per wave.
//
// Why this shape (measured, tools/exp/README_attention_gen2.md): at 32 keys per wave every 16 KiB of Q / dO fragments read from LDS feeds only 32
// MFMAs; the LDS traffic alone (15 GB per launch at B = 32, H = 12, N = 1568) is worth 121 us, three exposed LDS round trips per tile
// and two waves per SIMD contending for one matrix pipe do the rest.  With 64 keys per wave the same fragments feed 64 MFMAs -- but dK^T / dV^T of 64
// keys are 128 accumulator registers next to 64 for the resident K / V fragments and 64 for S / dP, more than the 256 architectural VGPRs.  hipcc
// picks ONE MFMA form per function (all accumulators in VGPRs, or all in AGPRs -- then the softmax pays a v_accvgpr_read per score), so the
// split is made by hand here: the file is compiled in the VGPR form (S and dP, which the vector ALU reads, stay in VGPRs), and the dV^T / dK^T products
// are issued from inline assembly with their accumulators pinned to AGPRs ("+a"), which only these MFMAs and the epilogue ever touch.
//
// Row constants ride in the operands: K' = -scale*log2(e) K and V' = -V are formed once per wave in registers, the S accumulators start from
// lse*log2(e) and the dP accumulators from delta, so p = exp2(-acc_S) (negation = input modifier) and -dS = p * acc_dP; dK takes the sign in its
// final scale.  Q / dO tiles (32 queries) and their statistics arrive by LDS-DMA through buffer descriptors two tiles ahead into a four-stage
// ring; the loop is unrolled over the stages so that every LDS address is a loop-invariant register plus an immediate.
// Reference math: model/modeling_slot.py:105-112 differentiated; layouts as attention.hip.
#include "common.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ int img_off(int row, int col) { return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + (col & 15) * 2; }
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// accumulate into an AGPR quad (inline assembly: the compiler's own MFMAs of this file are all VGPR-form).  Hazards: the accumulator is only ever
// touched by these MFMAs (same opcode, srcC = vDst: back-to-back is legal) until the epilogue, which pads before reading it.
__device__ __forceinline__ void mfma_agpr(f32x4& acc, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
struct TrFrag { u32x2 lo, hi; };
template <int IMM>
__device__ __forceinline__ u32x2 ds_read_tr_imm(unsigned addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
    return r;
}
__device__ __forceinline__ bf16x8 tr_assemble(const TrFrag& f) {
    const u32x4 r = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return *reinterpret_cast<const bf16x8*>(&r);
}
__device__ __forceinline__ void tr_fence4(TrFrag (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi) :: "memory");
}
__device__ __forceinline__ bf16x8 scale8(bf16x8 v, float s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)((float)v[j] * s);
    return r;
}
__device__ __forceinline__ float exp2_neg(float x) {
    float r;
    asm("v_exp_f32 %0, -%1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const bf16x2_ t = {(bf16)a, (bf16)b};
    return *reinterpret_cast<const unsigned*>(&t);
}
struct HeadMap { int blk, h, b; };
__device__ __forceinline__ HeadMap head_map(int nblk, int H, int B, bool xcd) {
    HeadMap m;
    if (!xcd) { m.blk = blockIdx.x; m.h = blockIdx.y; m.b = blockIdx.z; return m; }
    const int bid = blockIdx.x, x = bid & 7, slot = bid >> 3;
    const int hidx = (slot / nblk) * 8 + x;
    m.blk = slot - (slot / nblk) * nblk;
    m.h = hidx % H; m.b = hidx / H;
    return m;
}

enum { D3_STAGE = 4096 * 2 + 512, D3_NSTAGE = 4, KT = 4 };       // per stage: Q image (32 x 64 bf16) | dO image | lse[64] | delta[64]

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void mhsa_bwd_dkdv3_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                                          bf16* __restrict__ dqkv, int N, int H, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) char smem[D3_NSTAGE * D3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int KW = 16 * KT, KB = NW * KW, PIECES = 4 / NW;
    constexpr int PER_TILE = 2 * PIECES + 2;
    const HeadMap hm = head_map((N + KB - 1) / KB, H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const bf16* dobase = d_o + (int64_t)b * N * D + h * 64;
    const float* lse_bh = lse + ((int64_t)b * H + h) * N;
    const float* dl_bh = delta + ((int64_t)b * H + h) * N;
    const int key0 = hm.blk * KB + wave * KW;
    const bool active = key0 < N;
    const int nt = (N + 31) / 32;

    uint32_t voq[PIECES], voo[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = (wave * PIECES + i) * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = (((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1);
        voq[i] = (uint32_t)((row * (int)RS + chunk * 8) * 2);
        voo[i] = (uint32_t)((row * D + chunk * 8) * 2);
    }
    const uint32_t vos = (uint32_t)lane * 4;
    const int64_t qkv_rows = (int64_t)(xcd >> 16) * N - (int64_t)b * N;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, (int)min((int64_t)0x7fffffff, (qkv_rows * RS - h * 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dobase), 0, (int)min((int64_t)0x7fffffff, (qkv_rows * D - h * 64) * 2), 0x00020000);
    const int64_t stat_left = ((int64_t)(xcd >> 16) * H - ((int64_t)b * H + h)) * N * 4;
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse_bh), 0, (int)min((int64_t)0x7fffffff, stat_left), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dl_bh), 0, (int)min((int64_t)0x7fffffff, stat_left), 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned ro[2], tro[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ro[ks] = lds0 + img_off(c, 32 * ks + 8 * g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tro[dt] = lds0 + img_off(4 * g + (c >> 2), 16 * dt + 4 * (c & 3));
    const unsigned so = lds0 + 8192 + 16 * g;

    bf16x8 kreg[KT][2], vreg[KT][2];
    {
        const float ks_ = -scale * LOG2E;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = min(key0 + 16 * kt + c, N - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kreg[kt][ks] = scale8(*reinterpret_cast<const bf16x8*>(base + D + (int64_t)key * RS + 32 * ks + 8 * g), ks_);
                vreg[kt][ks] = scale8(*reinterpret_cast<const bf16x8*>(base + 2 * D + (int64_t)key * RS + 32 * ks + 8 * g), -1.0f);
            }
        }
    }
    f32x4 acc_dk[4][KT], acc_dv[4][KT];                      // AGPRs: [dt][kt], rows d = 16 dt + 4 g + r, column = key c of key tile kt
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dk[i][j][0])); asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dk[i][j][1]));
            asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dk[i][j][2])); asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dk[i][j][3]));
            asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dv[i][j][0])); asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dv[i][j][1]));
            asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dv[i][j][2])); asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc_dv[i][j][3]));
        }

    auto dma = [&](int t, int S) {
#if defined(__HIP_DEVICE_COMPILE__)
        char* st = smem + S * D3_STAGE;
        const int sq = t * 32 * (int)RS * 2, so_ = t * 32 * D * 2, ss = t * 128;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int r8 = (wave * PIECES + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_ptr)(st + r8 * 128), 16, voq[i], sq, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_void_ptr)(st + 4096 + r8 * 128), 16, voo[i], so_, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (lds_void_ptr)(st + 8192), 4, vos, ss, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_ptr)(st + 8192 + 256), 4, vos, ss, 0, 0);
#else
        (void)t; (void)S;
#endif
    };

#ifdef DEVIAS_ATTN_STAMPS
    unsigned long long st_wait = 0, st_bar = 0, st_p1 = 0, st_p2 = 0, st_p3 = 0;
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define STAMP(v)
#endif
    typedef std::integral_constant<bool, true> True_;
    typedef std::integral_constant<bool, false> False_;

    // one tile (ring stage S): S' / dP' (32 MFMAs, VGPR form) -> softmax arithmetic -> dV^T, dK^T (32 MFMAs, AGPR accumulators).  Key-tile major, so that
    // the scheduler can run the arithmetic of key tile kt under the MFMAs of kt + 1.  FULL: tiles i + 2, i + 3 exist and tile i is whole.
    auto body = [&](auto stage_tag, int i, auto full_tag) {
        constexpr int S = decltype(stage_tag)::value, S3 = (S + 3) & 3;
        constexpr bool FULL = decltype(full_tag)::value;
        typedef __attribute__((address_space(3))) const char* lp;
        // tile i has landed for this wave (tiles i + 1, i + 2 may still fly); after the barrier for every wave, and stage (i - 1) & 3 is free for tile i + 3
        STAMP(t0);
        if (FULL || i + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
        else if (i + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t1);
        __builtin_amdgcn_s_barrier();
        STAMP(t2);
        if (FULL || i + 3 < nt) dma(i + 3, S3);
        if (!active) return;
        // transposed fragments of dO (for dV^T) are requested first: they fly under the S / dP products
        TrFrag tfo[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            tfo[dt].lo = ds_read_tr_imm<S * D3_STAGE + 4096>(tro[dt]);
            tfo[dt].hi = ds_read_tr_imm<S * D3_STAGE + 4096 + 2048>(tro[dt]);
        }
        f32x4 l4[2], d4[2];
        bf16x8 qf[2][2], of[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            l4[qt] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>((lp)(size_t)(so + S * D3_STAGE + 64 * qt)) * LOG2E;
            d4[qt] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>((lp)(size_t)(so + S * D3_STAGE + 256 + 64 * qt));
            if constexpr (!FULL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) l4[qt][r] = (i * 32 + 16 * qt + 4 * g + r >= N) ? INFINITY : l4[qt][r];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[qt][ks] = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>((lp)(size_t)(ro[ks] + S * D3_STAGE + 2048 * qt));
                of[qt][ks] = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>((lp)(size_t)(ro[ks] + S * D3_STAGE + 4096 + 2048 * qt));
            }
        }
        bf16x8 pf[KT], dsf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 s[2], dp[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                s[qt] = mfma(qf[qt][1], kreg[kt][1], mfma(qf[qt][0], kreg[kt][0], l4[qt]));
                dp[qt] = mfma(of[qt][1], vreg[kt][1], mfma(of[qt][0], vreg[kt][0], d4[qt]));
            }
            u32x4 pw, dw;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float p0 = exp2_neg(s[qt][0]), p1 = exp2_neg(s[qt][1]), p2 = exp2_neg(s[qt][2]), p3 = exp2_neg(s[qt][3]);
                pw[2 * qt] = cvt_pk(p0, p1); pw[2 * qt + 1] = cvt_pk(p2, p3);
                dw[2 * qt] = cvt_pk(p0 * dp[qt][0], p1 * dp[qt][1]); dw[2 * qt + 1] = cvt_pk(p2 * dp[qt][2], p3 * dp[qt][3]);
            }
            pf[kt] = *reinterpret_cast<const bf16x8*>(&pw);
            dsf[kt] = *reinterpret_cast<const bf16x8*>(&dw);
        }
        tr_fence4(tfo);
        STAMP(t3);
        TrFrag tfq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            tfq[dt].lo = ds_read_tr_imm<S * D3_STAGE>(tro[dt]);
            tfq[dt].hi = ds_read_tr_imm<S * D3_STAGE + 2048>(tro[dt]);
        }
        STAMP(t3a);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dot = tr_assemble(tfo[dt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) mfma_agpr(acc_dv[dt][kt], dot, pf[kt]);
        }
        STAMP(t3b);
        tr_fence4(tfq);
        STAMP(t4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 qt_ = tr_assemble(tfq[dt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) mfma_agpr(acc_dk[dt][kt], qt_, dsf[kt]);
        }
#ifdef DEVIAS_ATTN_STAMPS
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t5 = __builtin_amdgcn_s_memtime();
        st_wait += t3a - t3; st_bar += t3b - t3a; st_p1 += t3 - t2; st_p2 += t4 - t3b; st_p3 += t5 - t4;
#endif
    };

    // ---- prologue: tiles 0, 1, 2 requested ----
    dma(0, 0);
    if (nt > 1) dma(1, 1);
    if (nt > 2) dma(2, 2);
    int i = 0;
    for (; i + 6 < nt && (i + 4) * 32 <= N; i += 4) {        // four FULL bodies: tiles up to i + 6 exist, tiles i .. i + 3 are whole
        body(std::integral_constant<int, 0>(), i, True_());
        body(std::integral_constant<int, 1>(), i + 1, True_());
        body(std::integral_constant<int, 2>(), i + 2, True_());
        body(std::integral_constant<int, 3>(), i + 3, True_());
    }
    for (; i < nt; ++i) {
        switch (i & 3) {
            case 0: body(std::integral_constant<int, 0>(), i, False_()); break;
            case 1: body(std::integral_constant<int, 1>(), i, False_()); break;
            case 2: body(std::integral_constant<int, 2>(), i, False_()); break;
            default: body(std::integral_constant<int, 3>(), i, False_()); break;
        }
    }
    if (!active) return;
#ifdef DEVIAS_ATTN_STAMPS
    if (lane == 0 && (blockIdx.x % 97) == 0) {            // overwrites a few delta entries of the scratch buffer: debug builds only
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(delta)) + ((blockIdx.x / 97) * NW + wave) * 8;
        dbg[0] = st_wait; dbg[1] = st_bar; dbg[2] = st_p1; dbg[3] = st_p2; dbg[4] = st_p3; dbg[5] = (unsigned long long)nt;
    }
#endif

    // ---- epilogue: the accumulators leave the AGPRs (padding: the last MFMAs must have written them back) ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = key0 + 16 * kt + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 vk, vv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(vk[r]) : "a"(acc_dk[dt][kt][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(vv[r]) : "a"(acc_dv[dt][kt][r]));
            }
            if (key < N) {
                bf16* row = dqkv + ((int64_t)b * N + key) * RS + h * 64 + 4 * g;
                store4(row + D + 16 * dt, vk * (-scale));          // dK = scale * dS^T Q, and the accumulated dS carries a minus sign
                store4(row + 2 * D + 16 * dt, vv);
            }
        }
    }
}

}  // namespace

// launcher used by devias_mhsa_bwd (attention.hip): nw = waves per workgroup (2 or 4), 64 keys per wave
int devias_launch_dkdv3(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv, int B, int N, int H, float scale, int xcd, int nw,
                        hipStream_t st) {
    const int kb = nw * 64, nblk = (N + kb - 1) / kb;
    const dim3 grid = (xcd & 1) ? dim3(nblk * H * B) : dim3(nblk, H, B);
    if (nw == 2) hipLaunchKernelGGL((mhsa_bwd_dkdv3_bf16_kernel<2>), grid, dim3(128), 0, st, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, xcd);
    else hipLaunchKernelGGL((mhsa_bwd_dkdv3_bf16_kernel<4>), grid, dim3(256), 0, st, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, xcd);
    return 0;
}
