// Second-generation bf16 attention backward kernels for gfx950: ONE wave per SIMD with the whole 512-entry register file, software-pipelined inside the wave.
//
// Why (measured, profiles/r3*): the first-generation dK/dV kernel (attention.hip: 2 waves per SIMD, 252 registers each, 64-query tiles) ran one
// tile-wave per ~2600 cycles for 1024 cycles of MFMA issue -- every group of 4-6 MFMAs waited for the LDS fragments it had just requested
// (no registers left to request them earlier), and the matrix phase and the softmax phase of a wave never overlapped; the second wave of the SIMD
// was in the same state.  Here a wave owns 64 keys (dK^T, dV^T of its keys: 128 accumulator registers; K, V fragments: 64), walks the queries in
// 32-row tiles, and its instruction stream for tile i interleaves
//     phase 1   S / dP MFMAs of tile i+1 (32)        with the softmax arithmetic of tile i, key tiles 0-1
//     phase 2a  dV / dK MFMAs of tile i, keys 0-1    with the softmax arithmetic of tile i, key tiles 2-3
//     phase 2b  dV / dK MFMAs of tile i, keys 2-3
// with every LDS fragment of a phase requested a phase ahead.  Q / dO tiles and their row statistics arrive by LDS-DMA two tiles ahead into a
// four-stage ring (one workgroup barrier per tile, counted vmcnt).  Row constants ride in the accumulators' initial value and in the register-resident
// operands: K is pre-multiplied by -scale*log2(e) and the S accumulators start from +lse*log2(e), V is negated and the dP accumulators start from
// +delta, so that p = exp2(-acc_S) (the negation is an input modifier) and -dS = p * acc_dP: two VALU operations and two packs per pair of
// scores instead of four (MI355X guide, "Row constants as the initial accumulator").  dK picks the sign up in its final scale.
// Reference math: model/modeling_slot.py:105-112 (softmax(q k^T * scale) v) differentiated; layouts as attention.hip.
#include "common.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// [rows][64 cols] bf16 image, 128-byte rows, 32-byte window XOR ((row >> 1) & 3): conflict-free for ds_read_b128 row fragments AND for
// ds_read_b64_tr_b16 transposed fragments (attention.hip: img_tr_off)
__device__ __forceinline__ int img_off(int row, int col) { return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + (col & 15) * 2; }

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// row fragment: lane holds tile[row = base + (lane & 15)][32 ks + 8 (lane >> 4) .. + 8]
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int base, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + img_off(base + (lane & 15), 32 * ks + 8 * (lane >> 4)));
}
// transposed fragment halves by inline asm (the builtin makes the compiler wait vmcnt(0) for the LDS-DMA in flight, see gemm.hip): lane holds
// tile[row = 16 (j >> 2) + 4 g + (j & 3)][col = cbase + (lane & 15)], j = 0..7, rows of ONE 32-row tile
struct TrFrag { u32x2 lo, hi; };
__device__ __forceinline__ u32x2 ds_read_tr_asm(const char* p) {
    u32x2 r;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ TrFrag frag_tr_issue(const char* img, int cbase, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int r0 = 4 * g + (c >> 2), col = cbase + 4 * (c & 3);
    TrFrag f;
    f.lo = ds_read_tr_asm(img + img_off(r0, col));
    f.hi = ds_read_tr_asm(img + img_off(r0 + 16, col));
    return f;
}
__device__ __forceinline__ bf16x8 tr_assemble(const TrFrag& f) {
    const u32x4 r = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return *reinterpret_cast<const bf16x8*>(&r);
}
// every outstanding LDS read of this wave has returned; the raw halves are operands so that nothing that reads them moves above the wait
__device__ __forceinline__ void tr_fence8(TrFrag (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi),
                 "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi), "+v"(f[6].lo), "+v"(f[6].hi), "+v"(f[7].lo), "+v"(f[7].hi) :: "memory");
}
__device__ __forceinline__ void tr_fence4(TrFrag (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi) :: "memory");
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
    return r;
}
__device__ __forceinline__ bf16x8 scale8(bf16x8 v, float s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)((float)v[j] * s);
    return r;
}
// exp2(-x): the negation is the instruction's input modifier
__device__ __forceinline__ float exp2_neg(float x) {
    float r;
    asm("v_exp_f32 %0, -%1" : "=v"(r) : "v"(x));
    return r;
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

enum { D2_STAGE = 4096 * 2 + 512, D2_NSTAGE = 4 };       // per stage: Q image (32 x 64 bf16) | dO image | lse[64] | delta[64] (fp32 as stored by forward / dQ; the first 32 of each are the tile's)

// transposed read with the stage / image / half as an IMMEDIATE offset: the per-lane address registers are loop invariants
template <int IMM>
__device__ __forceinline__ u32x2 ds_read_tr_imm(unsigned addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
    return r;
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const bf16x2_ t = {(bf16)a, (bf16)b};
    return *reinterpret_cast<const unsigned*>(&t);
}

// ======================================= backward dK, dV (bf16), generation 2 =====================================================
// workgroup = NW waves x 16 KT keys of one (batch, head); grid = ceil(N / (16 KT NW)) blocks per head.
// What bounds these kernels is the VECTOR INSTRUCTION COUNT (rocprofv3 PMC, profiles/r3_attn_pmc.txt): a 16x16x32 MFMA keeps the SIMD's issue
// port for 8 of its 16 cycles, so two plain vector instructions per MFMA are free and every further one costs its full 4-8 cycles -- the first
// generation issued 127 vector instructions per 32 MFMAs where the arithmetic needs ~56.  Hence: every LDS address is a loop-invariant register
// plus an immediate (the loop is unrolled over the four ring stages), every LDS-DMA source is a scalar base plus a loop-invariant lane offset,
// the row constants ride in the accumulators' initial values, and nothing is re-packed.
template <int NW, int KT, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void mhsa_bwd_dkdv2_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                                          bf16* __restrict__ dqkv, int N, int H, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) char smem[D2_NSTAGE * D2_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int KW = 16 * KT, KB = NW * KW, PIECES = 4 / NW;   // keys per wave / per workgroup; 1-KiB pieces of a 4-KiB image this wave stages
    constexpr int PER_TILE = 2 * PIECES + 2;                // LDS-DMA instructions per tile and wave (Q pieces, dO pieces, the two statistics lines)
    const HeadMap hm = head_map((N + KB - 1) / KB, H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const bf16* dobase = d_o + (int64_t)b * N * D + h * 64;
    const float* lse_bh = lse + ((int64_t)b * H + h) * N;
    const float* dl_bh = delta + ((int64_t)b * H + h) * N;
    const int key0 = hm.blk * KB + wave * KW;
    const bool active = key0 < N;                           // (a wave whose keys all lie beyond N only stages and synchronises)
    const int nt = (N + 31) / 32;

    // ---- loop-invariant per-lane offsets ----
    // LDS-DMA sources through buffer descriptors (buffer_load ... lds: descriptor + scalar tile offset + loop-invariant lane offset: no vector
    // arithmetic per tile, and rows past the end of the tensor read as zero).  Piece i of an image = rows 8 (wave PIECES + i) .. + 8; lane -> (row,
    // 16-byte slot), the image's swizzle applied to the slot.  Rows >= N of the last tile belong to the next batch entry (finite) or read as zero;
    // their probabilities are forced to zero (MASK below).
    uint32_t voq[PIECES], voo[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = (wave * PIECES + i) * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = (((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1);
        voq[i] = (uint32_t)((row * (int)RS + chunk * 8) * 2);
        voo[i] = (uint32_t)((row * D + chunk * 8) * 2);
    }
    const uint32_t vos = (uint32_t)lane * 4;
    const int64_t qkv_rows = (int64_t)(xcd >> 16) * N - (int64_t)b * N;      // rows from this batch entry to the end of the tensor
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, (int)min((int64_t)0x7fffffff, (qkv_rows * RS - h * 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dobase), 0, (int)min((int64_t)0x7fffffff, (qkv_rows * D - h * 64) * 2), 0x00020000);
    const int64_t stat_left = ((int64_t)(xcd >> 16) * H - ((int64_t)b * H + h)) * N * 4;
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse_bh), 0, (int)min((int64_t)0x7fffffff, stat_left), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dl_bh), 0, (int)min((int64_t)0x7fffffff, stat_left), 0x00020000);
    // LDS reads (byte offsets inside a stage's Q image; the dO image, the second 16 rows and the stage are immediates)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned ro[2], tro[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ro[ks] = lds0 + img_off(c, 32 * ks + 8 * g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tro[dt] = lds0 + img_off(4 * g + (c >> 2), 16 * dt + 4 * (c & 3));
    const unsigned so = lds0 + 8192 + 16 * g;                   // lse of rows 4 g .. (+ 64 QT); delta 256 bytes further

    // this wave's K and V rows as B operands (lane: key c of key tile kt, dims 32 ks + 8 g ..): K' = -scale*log2(e) K, V' = -V
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
    f32x4 acc_dk[4][KT], acc_dv[4][KT];                      // [dt][kt]: rows d = 16 dt + 4 g + r, column = key c of key tile kt
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) { acc_dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- staging: this wave's share of tile t into ring stage S (all by LDS-DMA: the only vector-memory operations of the loop, counted by hand)
    auto dma = [&](int t, int S) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass of hipcc does not know this builtin and, inside a template, silently drops the kernel's launch stub)
        char* st = smem + S * D2_STAGE;
        const int sq = t * 32 * (int)RS * 2, so_ = t * 32 * D * 2, ss = t * 128;        // scalar byte offsets of the tile
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int r8 = (wave * PIECES + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_ptr)(st + r8 * 128), 16, voq[i], sq, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_void_ptr)(st + 4096 + r8 * 128), 16, voo[i], so_, 0, 0);
        }
        // statistics: 64 rows of lse and of delta each (the tile's 32 and the next 32: a wave-instruction moves 64 x 4 bytes); every wave writes the
        // same bytes, which keeps the per-wave DMA count uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (lds_void_ptr)(st + 8192), 4, vos, ss, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_ptr)(st + 8192 + 256), 4, vos, ss, 0, 0);
#else
        (void)t; (void)S;
#endif
    };
    auto dma_any = [&](int t) { dma(t, t & (D2_NSTAGE - 1)); };

    // S / dP accumulators of one 32-query tile: [qt][kt], rows = queries 16 qt + 4 g + r, column = key c
    struct Acc { f32x4 s[2][KT], dp[2][KT]; };
    typedef std::integral_constant<bool, true> True_;
    typedef std::integral_constant<bool, false> False_;

    // phase 1, one 16-query half (QT) of the tile in stage S: S' = lse2 - (scale log2e) Q K^T and dP' = delta - dO V^T.  MASK: ragged last tile
    auto s_dp_half = [&](auto stage_tag, auto qt_tag, int t, Acc& a, auto mask_tag) {
        constexpr int S = decltype(stage_tag)::value, QT = decltype(qt_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value;
        typedef __attribute__((address_space(3))) const char* lp;
        f32x4 l4 = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>((lp)(size_t)(so + S * D2_STAGE + 64 * QT)) * LOG2E;
        const f32x4 d4 = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>((lp)(size_t)(so + S * D2_STAGE + 256 + 64 * QT));
        if constexpr (MASK) {                                            // rows past N get p = exp2(-inf) = 0 (branch-free: selects)
#pragma unroll
            for (int r = 0; r < 4; ++r) l4[r] = (t * 32 + 16 * QT + 4 * g + r >= N) ? INFINITY : l4[r];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                                 // (one 32-deep step at a time: 8 fragment registers in flight)
            const bf16x8 qf = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>((lp)(size_t)(ro[ks] + S * D2_STAGE + 2048 * QT));
            const bf16x8 of = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>((lp)(size_t)(ro[ks] + S * D2_STAGE + 4096 + 2048 * QT));
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if constexpr (ABL & 4) {
                    if (ks == 0) { a.s[QT][kt] = l4 + *reinterpret_cast<const f32x4*>(&qf); a.dp[QT][kt] = d4 + *reinterpret_cast<const f32x4*>(&of); }
                } else {
                a.s[QT][kt] = mfma(qf, kreg[kt][ks], ks == 0 ? l4 : a.s[QT][kt]);
                a.dp[QT][kt] = mfma(of, vreg[kt][ks], ks == 0 ? d4 : a.dp[QT][kt]);
                }
            }
        }
    };
    // softmax arithmetic of key tile kt: p = exp2(-S'), -dS = p * dP'; packed (pairs of consecutive rows of one 16-query half: no re-shuffling)
    // as the B operands of the dV^T / dK^T products
    auto soft = [&](const Acc& a, int kt, bf16x8& pf, bf16x8& dsf) {
        if constexpr (ABL & 2) {                                        // timing ablation: no arithmetic, the dependence on the accumulators kept
            pf = *reinterpret_cast<const bf16x8*>(&a.s[0][kt]); dsf = *reinterpret_cast<const bf16x8*>(&a.dp[1][kt]);
            return;
        }
        u32x4 pw, dw;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float p0 = exp2_neg(a.s[qt][kt][0]), p1 = exp2_neg(a.s[qt][kt][1]), p2 = exp2_neg(a.s[qt][kt][2]), p3 = exp2_neg(a.s[qt][kt][3]);
            pw[2 * qt] = cvt_pk(p0, p1); pw[2 * qt + 1] = cvt_pk(p2, p3);
            dw[2 * qt] = cvt_pk(p0 * a.dp[qt][kt][0], p1 * a.dp[qt][kt][1]); dw[2 * qt + 1] = cvt_pk(p2 * a.dp[qt][kt][2], p3 * a.dp[qt][kt][3]);
        }
        pf = *reinterpret_cast<const bf16x8*>(&pw);
        dsf = *reinterpret_cast<const bf16x8*>(&dw);
    };

    // ---- prologue: tiles 0, 1, 2 requested; S / dP of tile 0 ----
    dma_any(0);
    if (nt > 1) dma_any(1);
    if (nt > 2) dma_any(2);
    if (nt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) {                                          // same waits, barriers and DMA pieces as the active waves, no arithmetic
        for (int i = 0; i < nt; ++i) {
            if (i + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (i + 3 < nt) dma_any(i + 3);
        }
        return;
    }
    Acc A, Bc;
    s_dp_half(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), 0, A, True_());
    s_dp_half(std::integral_constant<int, 0>(), std::integral_constant<int, 1>(), 0, A, True_());

    // one iteration: tile i (ring stage S = i & 3) is finished (softmax, dV / dK) while tile i + 1's S / dP are formed.  FULL: tiles i + 1 .. i + 3
    // exist and are whole -- the body is then ONE basic block (the scheduler interleaves the softmax arithmetic with the MFMAs)
    auto body = [&](auto stage_tag, int i, Acc& cur, Acc& nxt, auto full_tag) {
        constexpr int S = decltype(stage_tag)::value, S1 = (S + 1) & 3, S3 = (S + 3) & 3;
        constexpr bool FULL = decltype(full_tag)::value;
        typedef std::integral_constant<int, S1> Next_;
        // tile i + 1 has landed for this wave (tile i + 2's DMA may still fly); after the barrier it has for every wave, and every wave is done with
        // stage (i - 1) & 3, which tile i + 3 now overwrites
        if constexpr (!(ABL & 1)) {
        if (FULL || i + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (FULL || i + 3 < nt) dma(i + 3, S3);
        }
        bf16x8 pf[KT], dsf[KT];
        constexpr int K1 = (KT + 1) / 2;                                 // key tiles whose dV / dK MFMAs form phase 2a
        // The sched_barriers fence the phases: without them the scheduler overlaps whole iterations and spills.
        __builtin_amdgcn_sched_barrier(0);
        // phase 1: S / dP of tile i + 1 (2 x 4 KT MFMAs) with the softmax arithmetic of tile i, key tiles 0 .. K1-1
        if constexpr (FULL) s_dp_half(Next_(), std::integral_constant<int, 0>(), i + 1, nxt, False_());
        else { if (i + 1 < nt) s_dp_half(Next_(), std::integral_constant<int, 0>(), i + 1, nxt, True_()); }
        soft(cur, 0, pf[0], dsf[0]);
        if constexpr (FULL) s_dp_half(Next_(), std::integral_constant<int, 1>(), i + 1, nxt, False_());
        else { if (i + 1 < nt) s_dp_half(Next_(), std::integral_constant<int, 1>(), i + 1, nxt, True_()); }
#pragma unroll
        for (int kt = 1; kt < K1; ++kt) soft(cur, kt, pf[kt], dsf[kt]);
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: transposed fragments of tile i, one operand at a time (16 registers in flight, not 32): dO^T under the rest of the softmax
        // arithmetic, then dV^T += dO^T P; Q^T under those MFMAs, then dK^T += Q^T dS
        TrFrag tfo[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            tfo[dt].lo = ds_read_tr_imm<S * D2_STAGE + 4096>(tro[dt]);
            tfo[dt].hi = ds_read_tr_imm<S * D2_STAGE + 4096 + 2048>(tro[dt]);
        }
#pragma unroll
        for (int kt = K1; kt < KT; ++kt) soft(cur, kt, pf[kt], dsf[kt]);
        tr_fence4(tfo);
        __builtin_amdgcn_sched_barrier(0);
        TrFrag tfq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            tfq[dt].lo = ds_read_tr_imm<S * D2_STAGE>(tro[dt]);
            tfq[dt].hi = ds_read_tr_imm<S * D2_STAGE + 2048>(tro[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dot = tr_assemble(tfo[dt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if constexpr (ABL & 8) acc_dv[dt][kt] += *reinterpret_cast<const f32x4*>(&dot) + *reinterpret_cast<const f32x4*>(&pf[kt]);
                else acc_dv[dt][kt] = mfma(dot, pf[kt], acc_dv[dt][kt]);
            }
        }
        tr_fence4(tfq);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 qt_ = tr_assemble(tfq[dt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if constexpr (ABL & 8) acc_dk[dt][kt] += *reinterpret_cast<const f32x4*>(&qt_) + *reinterpret_cast<const f32x4*>(&dsf[kt]);
                else acc_dk[dt][kt] = mfma(qt_, dsf[kt], acc_dk[dt][kt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int i = 0;
    for (; i + 6 < nt && (i + 5) * 32 <= N; i += 4) {       // four FULL bodies: tiles up to i + 6 exist, tiles up to i + 4 (whose S / dP they form) are whole
        body(std::integral_constant<int, 0>(), i, A, Bc, True_());
        body(std::integral_constant<int, 1>(), i + 1, Bc, A, True_());
        body(std::integral_constant<int, 2>(), i + 2, A, Bc, True_());
        body(std::integral_constant<int, 3>(), i + 3, Bc, A, True_());
    }
    for (; i < nt; ++i) {                                   // the last few tiles: same body with run-time guards (i & 3 is the stage, i & 1 the accumulator set)
        switch (i & 3) {
            case 0: body(std::integral_constant<int, 0>(), i, A, Bc, False_()); break;
            case 1: body(std::integral_constant<int, 1>(), i, Bc, A, False_()); break;
            case 2: body(std::integral_constant<int, 2>(), i, A, Bc, False_()); break;
            default: body(std::integral_constant<int, 3>(), i, Bc, A, False_()); break;
        }
    }

    // ---- epilogue: dV = acc_dv (P is positive: negating V only changed dP's sign), dK = scale * dS^T Q = -scale * acc_dk ----
    {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = key0 + 16 * kt + c;
            if (key < N) {
                bf16* row = dqkv + ((int64_t)b * N + key) * RS + h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4(row + D + 16 * dt, acc_dk[dt][kt] * (-scale));
                    store4(row + 2 * D + 16 * dt, acc_dv[dt][kt]);
                }
            }
        }
    }
}

}  // namespace

// launcher used by devias_mhsa_bwd (attention.hip): cfg = 10 * (waves per workgroup: 2 | 4) + (16-key tiles per wave: 3 | 4)
int devias_launch_dkdv2(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv, int B, int N, int H, float scale, int xcd, int cfg,
                        hipStream_t st) {
    const int abl = cfg / 100;
    cfg %= 100;
    const int nw = cfg / 10, kt = cfg % 10;
    const int kb = nw * 16 * kt, nblk = (N + kb - 1) / kb;
    const dim3 grid = (xcd & 1) ? dim3(nblk * H * B) : dim3(nblk, H, B);
#define DKDV2(NW, KT) hipLaunchKernelGGL((mhsa_bwd_dkdv2_bf16_kernel<NW, KT>), grid, dim3(NW * 64), 0, st, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, xcd)
#define DKDV2A(A) hipLaunchKernelGGL((mhsa_bwd_dkdv2_bf16_kernel<4, 2, A>), grid, dim3(256), 0, st, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, xcd)
    if (abl == 1) DKDV2A(1); else if (abl == 2) DKDV2A(2); else if (abl == 4) DKDV2A(4); else if (abl == 8) DKDV2A(8); else if (abl == 12) DKDV2A(12); else if (abl == 3) DKDV2A(3); else if (abl == 14) DKDV2A(14);
    else if (cfg == 22) DKDV2(2, 2);
    else DKDV2(4, 2);
#undef DKDV2A
#undef DKDV2
    return 0;
}
