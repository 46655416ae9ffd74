// Attention backward, dK / dV kernel in its one-wave-per-SIMD form (bf16, head dim 64, non-causal; model/modeling_slot.py:105-112 backward).
//
// Why: the two-waves-per-SIMD kernel (attention.hip: mhsa_bwd_dkdv_bf16_kernel) ran 13-44 % ABOVE the back-to-back sum of its own MFMA and VALU
// issue -- 44 % of its wave cycles parked on waits -- because nothing in a wave overlaps its matrix phase with its softmax phase and two phase-locked waves per
// SIMD do not either.  Here ONE wave owns a SIMD and the whole 512-register file, and the overlap is built into its instruction stream:
//
//   * a wave keeps dK^T and dV^T of its 64 keys in 128 accumulator registers (AGPRs named literally in inline asm, the technique of gemm256w_kernel) and its K / V
//     fragments in 64 more: the arch VGPRs carry only what flows -- S / dP tiles, Q / dO fragments, packed P / dS.  Workgroup = NW waves = NW x 64 keys of one
//     (batch, head): NW = 4 for the whole 256-key blocks, and the ragged rest of a head (N mod 256 keys) goes to a second, small launch instead of a 256-key workgroup
//     with idle waves (at N = 1568: 6.125 blocks -- a seventh workgroup with one busy wave would cost 14 %).  A rest of at most 64 keys (N = 1568: 32) is ONE wave
//     tile per head, and one wave walking all 49 slices is a latency chain (37 us for 384 lone waves on a 256-CU chip): QS = 2 / 4 puts that many waves on those keys, each with
//     its own rings and every QS-th slice, and adds their fp32 tiles through LDS in wave order at the end (QS chosen so that the launch stays one round of one wave per SIMD);
//   * Q / dO arrive in slices of 32 queries (one 4 KiB image each, read by rows for S / dP and transposed for dK^T / dV^T: a swizzle that is conflict-free for both
//     kinds of read at the 32x32x16 lane shapes -- SQ_LDS_BANK_CONFLICT = 0) by LDS-DMA into a ring of NST stages, counted vmcnt, one barrier per slice; the row
//     statistics of four slices ride in one more 1 KiB piece;
//   * v_mfma_f32_32x32x16_bf16 throughout: it holds the vector issue port for 8 of its 32 cycles, so ~5 single-issue instructions hide in every gap
//     (MI355X_MICROARCH.md, per-instruction constants).  Per slice 32 MFMAs in four groups of 8:
//         1  S / dP of (slice i, keys 32..63)     beside  softmax arithmetic of (slice i, keys 0..31); the LDS-DMA; row fragments + row constants of slice i + 1
//         2  dK^T / dV^T of (slice i - 1, 32..63) beside  the same arithmetic; the transposed fragments of slice i (each re-read behind the MFMA that consumed it)
//         3  S / dP of (slice i + 1, keys 0..31)  beside  softmax arithmetic of (slice i, keys 32..63)
//         4  dK^T / dV^T of (slice i, 0..31)      beside  the same
//     i.e. a unit's score tile is produced in one group, exponentiated over the next two and consumed by the group after: two score tiles (64 registers) in flight.
//   * per score one v_exp_f32, one v_mul_f32 and half a v_cvt_pk_bf16_f32 pair: the row constants ride in the MFMA's C operand (S accumulators start from
//     -lse * log2 e, dP accumulators from -delta) and K is pre-multiplied by scale * log2 e once per workgroup, so the MFMAs deliver S' = s - lse2 and dP - delta:
//     p = exp2(S'), p * (dP - delta) = dS / scale; dK is scaled by `scale` once at the end.
//   * epilogue: the tile goes through LDS so that every store instruction writes eight whole 128-byte rows (row-per-lane stores touch 64 lines each).
//
//   * PERSIST (option attn_dkdv = 2; not the default): one workgroup per CU walks the 256-key blocks w, w + grid, ...: the Q / dO ring never drains (the last
//     iterations of a block request the next block's first slices instead of re-reads; slice i of a block runs as unrolled copy (k0 + i) mod NST so that ring
//     stages stay compile-time constants), the next block's K / V rows are requested before the drain and the epilogue of the current one, the epilogue has its own
//     LDS behind the rings.  Bitwise equal to the one-block-per-workgroup form.  Measured: a block switch costs 9.7k cycles against 9.8k (prologue) + 4.1k (epilogue)
//     -- most of both is instruction issue (128 + 192 accumulator-register moves, 16 row-scattered K / V loads, the tile's trip through LDS), which nothing hides at
//     one wave per SIMD; the kernel alone 427 -> 420 us, the step +-0 (+0.06 ms, in-process A/B): DESIGN.md section 5, round 5.  Static lists also lose their
//     balance when another kernel holds CUs (the GEMMs' round-4 lesson), so the default stays one workgroup per block.
//
// The row statistics come pre-scaled and padded from the dQ kernel, which reads lse and computes delta anyway: stat [B, H, Npad / 32, 2, 32] fp32 (per 32-query
// slice -lse * log2 e | -delta; Npad = N rounded up to 32; -inf | 0 in the padding, so padded queries have p = 0 with no masking code).  Keys beyond N are computed on
// clamped rows and never stored.  No v_bias-gradient partials here: softmax rows sum to one, so sum_keys dV = sum_queries dO -- the caller takes that gradient from the
// column sums of dO (devias_mhsa_bwd_bias).  Deterministic (no atomics), bitwise run-to-run.
#include "attn1w.h"

namespace {

constexpr float LOG2E = LOG2E_1W;
enum { NSTATG = 4 /* statistics ring: groups of four slices, 1 KiB each */ };

// AGPR map (asm-owned: claimed once, never touched by the compiler -- audited in tests/test_build_cpu.py):
//   a[  0.. 63]  dV^T[db][kb]  (16 registers each, index db * 2 + kb)        a[128..159]  K fragments [kb][ks] (4 registers each), pre-multiplied by scale log2 e
//   a[ 64..127]  dK^T[db][kb]                                                a[160..191]  V fragments [kb][ks]
enum { A_DV = 0, A_DK = 64, A_K = 128, A_V = 160, A_END = 192 };
// Diagnostic builds only (tools/build_variant_file.sh <tag> attn_bwd1w -DDKDV_ABL=<mask>; results are then WRONG, the timing is what is read): leave out of the
// slice loop  1 = the LDS-DMA, 2 = the barrier, 4 = the softmax arithmetic, 8 = the row-fragment / row-constant reads, 16 = the transposed reads, 32 = the counted vmcnt;
// 64 = a v_mul in place of every v_exp, 128 = a v_perm in place of every v_cvt_pk, 256 = the S / dP MFMAs write (dummy) AGPRs instead of VGPRs,
// 512 = no wait for the transposed fragments in front of group 4, 1024 = no counted waits for the row fragments in group 3, 2048 = no epilogue (nothing is stored),
// 4096 = no K / V loads (zeros in the AGPRs: the MFMAs then draw less power and the clock rises -- this one measures the power management, not the loads)
#ifndef DKDV_ABL
#define DKDV_ABL 0
#endif
// S / dP: D (VGPRs) = A (VGPRs: a Q / dO row fragment) x B (AGPRs: a K / V fragment) + C (VGPRs: the row constants)
template <int BREG> __device__ __forceinline__ void mfma_init(f32x16& d, const bf16x8& a, const f32x16& c) {
    if constexpr (DKDV_ABL & 256) { asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" :: "v"(a), "i"(192 + (BREG >= 160 ? 16 : 0)), "i"(207 + (BREG >= 160 ? 16 : 0)), "i"(BREG), "i"(BREG + 3) : "a192", "a223"); asm volatile("" : "+v"(d) : "v"(c)); }
    else mfma_vab_init<BREG>(d, a, c);
}
template <int BREG> __device__ __forceinline__ void mfma_more(f32x16& d, const bf16x8& a) {
    if constexpr (DKDV_ABL & 256) { asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" :: "v"(a), "i"(192 + (BREG >= 160 ? 16 : 0)), "i"(207 + (BREG >= 160 ? 16 : 0)), "i"(BREG), "i"(BREG + 3)); asm volatile("" : "+v"(d)); }
    else mfma_vab_more<BREG>(d, a);
}
__device__ __forceinline__ void agpr_claim() { agpr_claim192(); }

// -DDKDV_STAMP builds: wave 0 of the first 4096 workgroups of the 256-key launch records the shader clock at [0] kernel entry, [1] loop entry, [2] loop exit, [3] kernel
// exit, [4] prefill issued, [5] K / V fragments and accumulators in place, [6] slice 0 landed (wait + barrier), [7] epilogue tile in LDS
// (devias_debug_dkdv_stamps reads them; each stamp drains the wave's LDS / scalar-memory counter, which is harmless at those four points)
__device__ unsigned long long g_dkdv_stamp[4096][8];
#ifndef DKDV_STAMP_ITEM
#define DKDV_STAMP_ITEM 0          // which of a persistent workgroup's items is stamped (0 = its first: with the prefill; 1 = its second: [0] is then the moment the first item is done)
#endif
#ifdef DKDV_STAMP
#define STAMP_REC(idx) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) g_dkdv_stamp[blockIdx.x][idx] = t_; }
#define STAMP(k) { if (NW == 4 && wave == 0 && blockIdx.x < 4096) { \
    if (stamp_item == DKDV_STAMP_ITEM && !(DKDV_STAMP_ITEM != 0 && ((k) == 0 || (k) == 4 || (k) == 5))) STAMP_REC(k) \
    else if (DKDV_STAMP_ITEM != 0 && stamp_item == DKDV_STAMP_ITEM - 1 && ((k) == 2 || (k) == 3)) STAMP_REC((k) + 2) } }      /* ITEM != 0: [4] / [5] = loop exit / stores issued of the item before */
#define STAMP_NEXT { ++stamp_item; if (NW == 4 && wave == 0 && blockIdx.x < 4096 && stamp_item == DKDV_STAMP_ITEM) STAMP_REC(0) }
#else
#define STAMP(k)
#define STAMP_NEXT
#endif

// =================================================================================================================================================================
// NW waves of 64 keys; the workgroup's first key is key_first + 64 NW * (its index within the head): the main launch covers the whole 256-key blocks (key_first = 0),
// the rest launch the ragged end (key_first = 256 * (N / 256), one workgroup per head).  NST = stages of the Q / dO ring.
template <int NW, int NST, bool PERSIST, int QS = 1>
__global__ __launch_bounds__(NW * QS * 64) void mhsa_bwd_dkdv1w_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o, const float* __restrict__ stat,
                                                                  bf16* __restrict__ dqkv, int N, int Npad, int H, int B, float scale, int xcd, int key_first, int nblk) {
    enum { PPW = 4 / NW /* 1 KiB pieces of each image per wave and slice */, DPS = 2 * PPW /* counted DMA instructions per wave and slice */,
           RING = NST * STAGE_BYTES, EPI = NW * 16384, EPI_OFF = PERSIST ? RING + NSTATG * 1024 : 0, WAVE_LDS = RING + NSTATG * 1024 /* QS > 1: rings per wave */,
           LDS_BYTES = QS > 1 ? (QS * WAVE_LDS > QS * 32768 ? QS * WAVE_LDS : QS * 32768) : PERSIST ? EPI_OFF + EPI : ((RING + NSTATG * 1024) > EPI ? (RING + NSTATG * 1024) : EPI) };
    static_assert(QS == 1 || ((QS == 2 || QS == 4) && NW == 1 && !PERSIST), "query split: one 64-key wave tile, two or four shares of the queries");
    // | Q / dO ring | statistics ring |   epilogue: one 16 KiB tile per wave (PERSIST: behind the rings, which stay live).  QS > 1: one such pair of rings PER WAVE
    // (the waves share their keys and split the query slices: slice j of wave w is slice QS j + w of the head), then one 32 KiB fp32 partial tile per wave
    __shared__ __attribute__((aligned(16))) char smem_all[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, r32 = lane & 31;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = QS > 1 ? 0 : wave_id;              // index of the wave's 64 keys within the workgroup
    const int wq = QS > 1 ? wave_id : 0;                // its share of the query slices
    char* const smem = smem_all + (QS > 1 ? wq * WAVE_LDS : 0);
    int stamp_item = 0; (void)stamp_item;
    STAMP(0)
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const int nsl_all = Npad >> 5;                      // query slices
    const int nsl = QS > 1 ? (nsl_all - wq + QS - 1) / QS : nsl_all;      // ... of this wave
    const int ngrp = (nsl_all + 3) >> 2;                // statistics pieces (groups of four slices) per head
    // item -> (batch, head, key block).  xcd & 1: all workgroups of one (batch, head) on one XCD (attention.hip head_map): its Q / dO rows stay in that L2.
    // PERSIST (whole 256-key blocks, xcd order only): the grid is a multiple of 8 workgroups, workgroup w takes items w, w + grid, ... -- an item stays on the
    // XCD of its head, and a head's blocks run at the same time on neighbouring workgroups
    const int nitems = PERSIST ? nblk * H * B : 0, istride = PERSIST ? (int)gridDim.x : 0;
    int item = blockIdx.x;
    struct Item { int b, h, blk; };
    auto decode = [&](int it) -> Item {
        Item r;
        if (PERSIST || (xcd & 1)) {
            const int x = it & 7, slot = it >> 3, hidx = (slot / nblk) * 8 + x;
            r.blk = slot - (slot / nblk) * nblk; r.h = hidx % H; r.b = hidx / H;
        } else { r.blk = blockIdx.x; r.h = blockIdx.y; r.b = blockIdx.z; }
        return r;
    };
    Item cur = decode(item);
    int b = cur.b, h = cur.h;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    int key0 = key_first + cur.blk * (64 * NW) + wave * 64;
    const bool active = PERSIST || key0 < N;            // (a wave without a valid key only stages and synchronises; the 256-key blocks have none)

    // ---- LDS-DMA: per wave and slice PPW 1 KiB pieces of the Q image (piece p = rows 8 p .. + 8), PPW of the dO image; with every fourth slice the 1 KiB of row
    // statistics of four slices (every wave writes the same bytes: the duplicate costs less than a wave-dependent vmcnt count) ----------------------------------------
    // PERSIST: the descriptors span the whole tensors and an item is a scalar byte offset (offq / offo / offs; *_n: the NEXT item's, whose first NST - 1 slices the
    // last iterations of this item's loop request in place of the harmless re-reads, so that the ring never drains between items)
    __amdgpu_buffer_rsrc_t rs_q, rs_o, rs_s;
    uint32_t vo_q[PPW], vo_o[PPW];
    int offq = 0, offo = 0, offs = 0, offq_n = 0, offo_n = 0, offs_n = 0;
    int gb = 0;                                               // statistics ring: piece g of this item sits in slot (gb + g) mod NSTATG
    auto item_offsets = [&](const Item& t, int& oq, int& oo, int& os) {
        oq = (int)((((int64_t)t.b * N) * RS + t.h * 64) * 2); oo = (int)((((int64_t)t.b * N) * D + t.h * 64) * 2); os = (int)((((int64_t)t.b * H + t.h) * 2 * Npad) * 4);
    };
    {
        if constexpr (PERSIST) {
            const int64_t bq = (int64_t)B * N * RS * 2, bo = (int64_t)B * N * D * 2, bs = (int64_t)B * H * 2 * Npad * 4;      // (< 2^31: the launcher checks)
            rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(qkv), 0, (int)bq, 0x00020000);
            rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(d_o), 0, (int)bo, 0x00020000);
            rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stat), 0, (int)bs, 0x00020000);
            item_offsets(cur, offq, offo, offs);
        } else {
            const bf16* dobase = d_o + (int64_t)b * N * D + h * 64;
            const float* stbase = stat + ((int64_t)b * H + h) * 2 * Npad;
            const int64_t left_q = ((int64_t)B - b) * N * RS - h * 64, left_o = ((int64_t)B - b) * N * D - h * 64;
            const int64_t bq = left_q * 2, bo = left_o * 2;
            rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, (int)(bq < 0x7fffffff ? bq : 0x7fffffff), 0x00020000);
            rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dobase), 0, (int)(bo < 0x7fffffff ? bo : 0x7fffffff), 0x00020000);
            rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stbase), 0, 2 * Npad * 4, 0x00020000);
        }
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int row = (wave + NW * k) * 8 + (lane >> 3), c = (lane & 7) ^ swz(row);
            vo_q[k] = (uint32_t)((row * (int)RS + c * 8) * 2);
            vo_o[k] = (uint32_t)((row * D + c * 8) * 2);
        }
    }
    const uint32_t vo_s = (uint32_t)(lane * 16);
    const int qstride = (int)RS * 64, ostride = D * 64;      // bytes per slice of 32 rows
    // slice -> stage `stage_off` (a byte offset into the ring, a compile-time constant in the loop).  A slice index >= nsl: PERSIST, the next item's slice - nsl;
    // otherwise a harmless re-read (it keeps the per-iteration DMA count, which the counted vmcnt relies on, constant)
    auto dma_q = [&](int slice, int stage_off, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
        const bool nx = PERSIST && slice >= nsl;
        const int sl = QS * max(min(nx ? slice - nsl : slice, nsl - 1), 0) + wq;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void_ptr)(smem + stage_off + (wave + NW * k) * 1024), 16, vo_q[k], (nx ? offq_n : offq) + sl * qstride, 0, 0);
#else
        (void)slice; (void)stage_off; (void)k;
#endif
    };
    auto dma_o = [&](int slice, int stage_off, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
        const bool nx = PERSIST && slice >= nsl;
        const int sl = QS * max(min(nx ? slice - nsl : slice, nsl - 1), 0) + wq;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (lds_void_ptr)(smem + stage_off + IMG_BYTES + (wave + NW * k) * 1024), 16, vo_o[k], (nx ? offo_n : offo) + sl * ostride, 0, 0);
#else
        (void)slice; (void)stage_off; (void)k;
#endif
    };
    // the statistics of slices 4 g .. 4 g + 3 (1 KiB, contiguous in `stat`), issued in front of slice 4 g's pieces; beyond Npad the buffer reads zero (PERSIST: the
    // next head's numbers, which nothing consumes).  At most four pieces are live at a time, also across an item boundary (the last piece of an item may be partial)
    auto dma_s = [&](int slice) {
#if defined(__HIP_DEVICE_COMPILE__)
        const bool nx = PERSIST && slice >= nsl;
        const int sl = nx ? slice - nsl : slice;
        if (((QS * sl) & 3) == 0) {                          // (QS > 1: a wave's slice j is slice QS j + wq of the head: piece (QS j) >> 2)
            const int g = (QS * sl) >> 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lds_void_ptr)(smem + RING + ((gb + (nx ? ngrp : 0) + g) & (NSTATG - 1)) * 1024), 16, vo_s, (nx ? offs_n : offs) + g * 1024, 0, 0);
        }
#else
        (void)slice;
#endif
    };
    // (Tried: the 64 K rows and 64 V rows of a wave as whole 128-byte rows by LDS-DMA into a staging area, fragments by ds_read_b128, instead of the sixteen row-per-lane
    // loads below, which touch 32 cache lines per instruction for 32 bytes each: bitwise equal, 808.9 us against 807.4 for the whole backward -- the wave waits for the
    // rings' prefill either way.  An ablation build WITHOUT the loads is 87 us faster, but it multiplies zeros: the matrix cores draw less and the clock rises --
    // ablations that change the DATA measure the power management, not the code: profiles/r5_dkdv1w_development.txt.)
    // ---- K / V fragments: requested FIRST -- vmcnt completes in issue order, so behind the ring's prefill their wait would be a wait for the whole prefill
    // (56 KiB at the ~11 B/clk a CU gets from a cold start: measured 11.9k cycles of prologue per workgroup, 14 % of its life) ----------------------------------------
    bf16x8 kv_[8], vv_[8];
    auto load_kv = [&](const bf16* ibase, int ikey0) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int kb = f >> 2, ks = f & 3;
            const int key = min(ikey0 + 32 * kb + r32, N - 1);
            kv_[f] = *reinterpret_cast<const bf16x8*>(ibase + D + (int64_t)key * RS + 16 * ks + 8 * hi);
            vv_[f] = *reinterpret_cast<const bf16x8*>(ibase + 2 * D + (int64_t)key * RS + 16 * ks + 8 * hi);
        }
    };
    if constexpr (DKDV_ABL & 4096) { _Pragma("unroll") for (int f = 0; f < 8; ++f) { kv_[f] = bf16x8{}; vv_[f] = bf16x8{}; } } else
    if (active) load_kv(base, key0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        dma_s(s);
#pragma unroll
        for (int k = 0; k < PPW; ++k) { dma_q(s, s * STAGE_BYTES, k); dma_o(s, s * STAGE_BYTES, k); }
    }
    STAMP(4)
    // Counted waits: all but the DPS * n youngest LDS-DMA instructions of this wave have landed.  (A statistics piece among the youngest makes the wait stricter by
    // one instruction, never laxer; the piece a slice needs is issued in front of that slice's own pieces or earlier.)
#define WAIT_SLICES_BUT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (n)) : "memory");

    if (!active) {
        WAIT_SLICES_BUT(NST - 2)
        if constexpr (NW > 1) __builtin_amdgcn_s_barrier();
        int st_d = (NST - 1) * STAGE_BYTES;
        for (int i = 0; i < nsl; ++i) {
            WAIT_SLICES_BUT(NST - 3)
            if constexpr (NW > 1) __builtin_amdgcn_s_barrier();
            dma_s(i + NST - 1);
#pragma unroll
            for (int k = 0; k < PPW; ++k) { dma_q(i + NST - 1, st_d, k); dma_o(i + NST - 1, st_d, k); }
            st_d += STAGE_BYTES; if (st_d == RING) st_d = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (NW > 1) __builtin_amdgcn_s_barrier();       // (the epilogue's barrier)
        return;
    }

    // ---- K / V fragments -> AGPRs (B operands: lane = key r32 of key block kb, d = 16 ks + 8 hi .. + 8); accumulators = 0 ------------------------------------------
    agpr_claim();
    auto fill_kv = [&]() {
        const float ksc = (xcd & 2) ? 1.0f : scale * LOG2E;      // (xcd bit 1 = ATTN_QPRE, attention.hip: q already holds q * scale * log2 e -- the scores are then the forward's own products)
        sfor<8>([&](auto I) {
            constexpr int f = decltype(I)::value;
            sfor<4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                agpr_write1<A_K + f * 4 + j>(cvt_pk_bf16((float)kv_[f][2 * j] * ksc, (float)kv_[f][2 * j + 1] * ksc));
                agpr_write1<A_V + f * 4 + j>(cvt_pk_bf16((float)vv_[f][2 * j], (float)vv_[f][2 * j + 1]));
            });
        });
        sfor<128>([&](auto I) { agpr_zero1<decltype(I)::value>(); });
    };
    fill_kv();

    STAMP(5)
    // ---- loop-invariant LDS byte offsets (the stage offset is added once per slice: nine vector adds) ----------------------------------------------------------------
    int row_a[4], tr_a[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_a[ks] = img_off(r32, 2 * ks + hi);
    {
        const int g16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < 2; ++db) tr_a[t][db] = img_off(8 * t + 4 * hi + q, 4 * db + 2 * g16 + (p >> 1)) + 8 * (p & 1);     // (+ 2048 s, + IMG_BYTES for dO)
    }
    const int st_a = RING + 16 * hi + (QS > 1 ? wq * 256 : 0);        // row constants of a slice: floats 8 jj + 4 hi .. + 4 of its 256 B (+ 128: delta)
    constexpr int SC_STEP = QS * 256;                                  // from a slice's constants to the next one's
    const lds_cptr lbase = (lds_cptr)smem;
    // the asm reads take LDS byte addresses (the kernel's one LDS array starts at 0): the same offsets plus the base of this wave's rings
    int row_w[4], tr_w[2][2];
    {
        const int wbase = QS > 1 ? wq * WAVE_LDS : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) row_w[ks] = row_a[ks] + wbase;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < 2; ++db) tr_w[t][db] = tr_a[t][db] + wbase;
    }

    // ---- pipeline state ----------------------------------------------------------------------------------------------------------------------------------------
    f32x16 S0, P0, S1, P1;                  // score / dP tiles of the two units in flight (unit = slice x 32-key block kb)
    f32x16 cL, cD;                          // the slice's row constants in accumulator layout: -lse2 / -delta of query (j & 3) + 8 (j >> 2) + 4 hi
    bf16x8 rq[4], ro[4];                    // Q / dO row fragments of the slice (A operands of S / dP)
    u32x2 tq[2][2][2], to[2][2][2];         // transposed Q / dO fragments [s][db][t] (A operands of dK^T / dV^T), as the two 8-byte reads they arrive in
    unsigned pw0[8], dw0[8], pw1[8], dw1[8];    // packed P / dS of units kb = 0 / 1 (register pairs (2 m, 2 m + 1) of the tile -> one word)
    float ev[2][16], dvv[2];                // (per-score exponentials of the two units in flight: registers, every index is a constant)
    auto reset_state = [&]() {              // per item: slice 0 finishes "score 15 of unit (-1, 1)" (0 * 0) and adds "unit (-1, 1)" (zero operands) to the accumulators
#pragma unroll
        for (int m = 0; m < 8; ++m) { pw0[m] = 0u; dw0[m] = 0u; pw1[m] = 0u; dw1[m] = 0u; }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int t = 0; t < 2; ++t) { tq[s][db][t] = u32x2{0u, 0u}; to[s][db][t] = u32x2{0u, 0u}; }
#pragma unroll
        for (int j = 0; j < 16; ++j) { P1[j] = 0.f; if constexpr (DKDV_ABL & 256) { S0[j] = 0.f; P0[j] = 0.f; S1[j] = 0.f; } }
#pragma unroll
        for (int g = 0; g < 16; ++g) { ev[0][g] = 0.f; ev[1][g] = 0.f; }
        dvv[0] = 0.f; dvv[1] = 0.f;
    };

#define LD_ROW_(ptr, img) (*reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((ptr) + (img)))
#define LD_C4_(ptr, which, jj) (*reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((ptr) + 128 * (which) + 32 * (jj)))
#define SET_C4(V, jj, X) { const f32x4 x_ = (X); V[4 * (jj)] = x_[0]; V[4 * (jj) + 1] = x_[1]; V[4 * (jj) + 2] = x_[2]; V[4 * (jj) + 3] = x_[3]; }
    auto frag = [&](const u32x2& lo, const u32x2& hi2) -> bf16x8 { return frag_of(lo, hi2); };
    auto pfrag = [&](const unsigned (&w)[8], int s) -> bf16x8 {
        const u32x4 v = {w[4 * s], w[4 * s + 1], w[4 * s + 2], w[4 * s + 3]};
        return *reinterpret_cast<const bf16x8*>(&v);
    };
    // softmax arithmetic, one score per MFMA gap, software-pipelined by one gap so that no v_mul waits on the v_exp in front of it: FIN finishes the score started one
    // gap earlier (p * dPn; every second score packs a register pair), EXPG starts score g (p = exp2(-Sn))
#define EXP_(x) ((DKDV_ABL & 64) ? (x) * 0.25f : fast_exp2(x))
#define PK_(a, b) ((DKDV_ABL & 128) ? __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u) : cvt_pk_bf16(a, b))
#define EXPG(S_, g) if constexpr (!(DKDV_ABL & 4)) { ev[&S_ == &S1][g] = EXP_(S_[g]); }
#define FIN(P_, PW_, DW_, g) if constexpr (!(DKDV_ABL & 4)) { constexpr int u_ = 0; (void)u_; const int uu = (&P_ == &P1); if ((g) & 1) { const float d1_ = ev[uu][g] * P_[g]; PW_[(g) >> 1] = PK_(ev[uu][((g) | 1) - 1], ev[uu][g]); DW_[(g) >> 1] = PK_(dvv[uu], d1_); } else dvv[uu] = ev[uu][g] * P_[g]; }
    // which score's arithmetic sits in which MFMA gap: one score per gap, each finished one gap behind its v_exp.  (Other placements -- none beside the MFMAs that
    // write VGPRs and two per other gap, one v_exp beside each of the former, ... -- measure the same within noise: profiles/r5_dkdv1w_development.txt)
#define SM_0 FIN(P1, pw1, dw1, 15) EXPG(S0, 0)
#define SM_1 FIN(P0, pw0, dw0, 0) EXPG(S0, 1)
#define SM_2 FIN(P0, pw0, dw0, 1) EXPG(S0, 2)
#define SM_3 FIN(P0, pw0, dw0, 2) EXPG(S0, 3)
#define SM_4 FIN(P0, pw0, dw0, 3) EXPG(S0, 4)
#define SM_5 FIN(P0, pw0, dw0, 4) EXPG(S0, 5)
#define SM_6 FIN(P0, pw0, dw0, 5) EXPG(S0, 6)
#define SM_7 FIN(P0, pw0, dw0, 6) EXPG(S0, 7)
#define SM_8 FIN(P0, pw0, dw0, 7) EXPG(S0, 8)
#define SM_9 FIN(P0, pw0, dw0, 8) EXPG(S0, 9)
#define SM_10 FIN(P0, pw0, dw0, 9) EXPG(S0, 10)
#define SM_11 FIN(P0, pw0, dw0, 10) EXPG(S0, 11)
#define SM_12 FIN(P0, pw0, dw0, 11) EXPG(S0, 12)
#define SM_13 FIN(P0, pw0, dw0, 12) EXPG(S0, 13)
#define SM_14 FIN(P0, pw0, dw0, 13) EXPG(S0, 14)
#define SM_15 FIN(P0, pw0, dw0, 14) EXPG(S0, 15)
#define SM_16 FIN(P0, pw0, dw0, 15) EXPG(S1, 0)
#define SM_17 FIN(P1, pw1, dw1, 0) EXPG(S1, 1)
#define SM_18 FIN(P1, pw1, dw1, 1) EXPG(S1, 2)
#define SM_19 FIN(P1, pw1, dw1, 2) EXPG(S1, 3)
#define SM_20 FIN(P1, pw1, dw1, 3) EXPG(S1, 4)
#define SM_21 FIN(P1, pw1, dw1, 4) EXPG(S1, 5)
#define SM_22 FIN(P1, pw1, dw1, 5) EXPG(S1, 6)
#define SM_23 FIN(P1, pw1, dw1, 6) EXPG(S1, 7)
#define SM_24 FIN(P1, pw1, dw1, 7) EXPG(S1, 8)
#define SM_25 FIN(P1, pw1, dw1, 8) EXPG(S1, 9)
#define SM_26 FIN(P1, pw1, dw1, 9) EXPG(S1, 10)
#define SM_27 FIN(P1, pw1, dw1, 10) EXPG(S1, 11)
#define SM_28 FIN(P1, pw1, dw1, 11) EXPG(S1, 12)
#define SM_29 FIN(P1, pw1, dw1, 12) EXPG(S1, 13)
#define SM_30 FIN(P1, pw1, dw1, 13) EXPG(S1, 14)
#define SM_31 FIN(P1, pw1, dw1, 14) EXPG(S1, 15)
#define SM_TAIL FIN(P1, pw1, dw1, 15)
    // an asm MFMA reads its C operand while it runs; the compiler, which does not see the MFMA, would hand a dead C's registers to the next temporaries
    // (v_exp results) and overwrite them under it: KEEP extends the operand's life past the hazard window (no instruction)
#define KEEP(V) asm volatile("" ::"v"(V));
#define LDL_C4(V, jj, ptr, which) if constexpr (!(DKDV_ABL & 8)) SET_C4(V, jj, LD_C4_(ptr, which, jj))
#define LDL_ROW(dst, ptr, img) if constexpr (!(DKDV_ABL & 8)) dst = LD_ROW_(ptr, img);
#define TRL(dst, OFF, a) if constexpr (!(DKDV_ABL & 16)) dst = lds_tr_off<OFF>(a);
#define DMA_Q(k) if constexpr (!(DKDV_ABL & 1) && (k) < PPW) dma_q(i + NST - 1, st_d, (k) < PPW ? (k) : 0);
#define DMA_O(k) if constexpr (!(DKDV_ABL & 1) && (k) < PPW) dma_o(i + NST - 1, st_d, (k) < PPW ? (k) : 0);

    // ---- per item: slice 0's row fragments and constants, S / dP of unit (0, 0).  k0s = byte offset of the ring stage that holds the item's slice 0 -------------
    int sc_n = 0;                                                            // offset of slice i + 1's constants in the statistics ring
    auto item_prologue = [&](bool first, int k0s) {
        agpr_claim();                        // (no compiler value may sit in a[0..191] across an item: it hoists the epilogue's addresses out of the item loop and parks them wherever it believes free)
        reset_state();
        if (first) { WAIT_SLICES_BUT(NST - 2) }                              // the prefill's slices 0 and 1
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");               // everything older than the previous item's 16 stores: its trailing DMA = this item's slices 0 .. NST - 2
        if constexpr (NW > 1) __builtin_amdgcn_s_barrier();
        STAMP(6)
        const int sc0 = (gb & (NSTATG - 1)) * 1024;
        {
            const lds_cptr cp = lbase + st_a + sc0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { SET_C4(cL, jj, LD_C4_(cp, 0, jj)); SET_C4(cD, jj, LD_C4_(cp, 1, jj)); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { rq[ks] = LD_ROW_(lbase + k0s + row_a[ks], 0); ro[ks] = LD_ROW_(lbase + k0s + row_a[ks], IMG_BYTES); }
        }
        SB
        mfma_init<A_K + 0>(S0, rq[0], cL);  mfma_init<A_V + 0>(P0, ro[0], cD);
        mfma_more<A_K + 4>(S0, rq[1]);      mfma_more<A_V + 4>(P0, ro[1]);
        mfma_more<A_K + 8>(S0, rq[2]);      mfma_more<A_V + 8>(P0, ro[2]);
        mfma_more<A_K + 12>(S0, rq[3]);     mfma_more<A_V + 12>(P0, ro[3]);
        SB
        sc_n = (sc0 + SC_STEP) & (NSTATG * 1024 - 1);
        STAMP(1)
    };
    // LDS reads of the loop.  The row constants (cL / cD: sixteen registers each, filled four at a time) are plain loads: the compiler, which counts only its own LDS
    // instructions, waits for them at CWAIT -- placed BEFORE the first asm read of the iteration, so that its lgkmcnt(0) never sits behind a young read it does not
    // see.  Row fragments (RLD) and transposed fragments (TRL) are asm reads with hand-counted waits: they return in issue order, and lgkmcnt is a 4-bit counter --
    // LWAIT(n) = all but the n <= 15 youngest LDS reads are back.  Issue order per iteration: rq0 ro0 rq1 | ro1 T0 | rq2 T1 | ro2 T2 | rq3 T3 | ro3 T4 | T5 | T6 | T7
    // (Tk = the two reads of transposed fragment k), one `|` per MFMA gap from gap 6 on; every wait below is for a read issued at least five gaps earlier.
#define RLD(dst, a, OFF) if constexpr (!(DKDV_ABL & 8)) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(a), "i"(OFF) : "memory");
#define LWAIT(n, x) if constexpr (!(DKDV_ABL & 1024)) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(n) : "memory");
#define CWAIT asm volatile("" ::"v"(cL), "v"(cD));
    // One slice per call; NST calls per trip of the loop below, the ring stage of slice i a compile-time constant in each: the stage offsets of slice i (transposed
    // reads), slice i + 1 (row reads) and of the slice this iteration's DMA fills fold into the instructions' immediates -- with run-time stages the loop carried nine
    // vector adds and the scalar bookkeeping of three ring cursors per slice, in a wave whose issue slots are the bound.
    auto slice_step = [&](const int i, auto STG) {
        constexpr int st_i = decltype(STG)::value * STAGE_BYTES, st_n = ((decltype(STG)::value + 1) % NST) * STAGE_BYTES, st_d = ((decltype(STG)::value + NST - 1) % NST) * STAGE_BYTES;
        // slice i + 1 has landed for this wave ... and for every wave; everyone is done with slice i - 1's stage
        if constexpr (!(DKDV_ABL & 32)) WAIT_SLICES_BUT(NST - 3)
        if constexpr (NW > 1 && !(DKDV_ABL & 2)) __builtin_amdgcn_s_barrier();
        const unsigned rp0 = (unsigned)row_w[0], rp1 = (unsigned)row_w[1], rp2 = (unsigned)row_w[2], rp3 = (unsigned)row_w[3];      // (+ st_n: an immediate of the read)
        const lds_cptr cp = lbase + sc_n + st_a;
        const unsigned t00 = (unsigned)tr_w[0][0], t10 = (unsigned)tr_w[1][0], t01 = (unsigned)tr_w[0][1], t11 = (unsigned)tr_w[1][1];   // (+ st_i)
        SB
        // ---- group 1: S / dP of unit (i, 1)   || softmax of unit (i, 0), scores 0..7; the DMA of slice i + NST - 1; the row constants of slice i + 1
        mfma_init<A_K + 16>(S1, rq[0], cL);  SM_0  if constexpr (!(DKDV_ABL & 1)) dma_s(i + NST - 1);  DMA_Q(0)  SB
        mfma_init<A_V + 16>(P1, ro[0], cD);  SM_1  DMA_O(0)  SB
        mfma_more<A_K + 20>(S1, rq[1]);      SM_2  DMA_Q(1)  KEEP(cL)  LDL_C4(cL, 0, cp, 0) LDL_C4(cL, 1, cp, 0) LDL_C4(cL, 2, cp, 0) LDL_C4(cL, 3, cp, 0)  SB
        mfma_more<A_V + 20>(P1, ro[1]);      SM_3  DMA_O(1)  KEEP(cD)  LDL_C4(cD, 0, cp, 1) LDL_C4(cD, 1, cp, 1) LDL_C4(cD, 2, cp, 1) LDL_C4(cD, 3, cp, 1)  SB
        mfma_more<A_K + 24>(S1, rq[2]);      SM_4  DMA_Q(2)  SB
        mfma_more<A_V + 24>(P1, ro[2]);      SM_5  DMA_O(2)  SB
        mfma_more<A_K + 28>(S1, rq[3]);      SM_6  DMA_Q(3)  CWAIT  RLD(rq[0], rp0, st_n + 0)  SB
        mfma_more<A_V + 28>(P1, ro[3]);      SM_7  DMA_O(3)  RLD(ro[0], rp0, st_n + IMG_BYTES)  SB
        // ---- group 2: dV^T / dK^T of unit (i - 1, 1)   || softmax of unit (i, 0), scores 8..15; the row fragments of slice i + 1 (each behind the MFMA that read
        //      the old one) and the transposed fragments of slice i (each behind the MFMA that consumed its previous contents, in the order group 4 consumes them)
        {
            const bf16x8 b0 = pfrag(pw1, 0), e0 = pfrag(dw1, 0), b1 = pfrag(pw1, 1), e1 = pfrag(dw1, 1);
            mfma_agpr<A_DV + 16>(frag(to[0][0][0], to[0][0][1]), b0);  SM_8   RLD(rq[1], rp1, st_n + 0)  SB
            mfma_agpr<A_DK + 16>(frag(tq[0][0][0], tq[0][0][1]), e0);  SM_9   RLD(ro[1], rp1, st_n + IMG_BYTES)  TRL(to[0][0][0], st_i + IMG_BYTES, t00)  TRL(to[0][0][1], st_i + IMG_BYTES, t10)  SB
            mfma_agpr<A_DV + 48>(frag(to[0][1][0], to[0][1][1]), b0);  SM_10  RLD(rq[2], rp2, st_n + 0)          TRL(tq[0][0][0], st_i + 0, t00)          TRL(tq[0][0][1], st_i + 0, t10)          SB
            mfma_agpr<A_DK + 48>(frag(tq[0][1][0], tq[0][1][1]), e0);  SM_11  RLD(ro[2], rp2, st_n + IMG_BYTES)  TRL(to[0][1][0], st_i + IMG_BYTES, t01)  TRL(to[0][1][1], st_i + IMG_BYTES, t11)  SB
            mfma_agpr<A_DV + 16>(frag(to[1][0][0], to[1][0][1]), b1);  SM_12  RLD(rq[3], rp3, st_n + 0)          TRL(tq[0][1][0], st_i + 0, t01)          TRL(tq[0][1][1], st_i + 0, t11)          SB
            mfma_agpr<A_DK + 16>(frag(tq[1][0][0], tq[1][0][1]), e1);  SM_13  RLD(ro[3], rp3, st_n + IMG_BYTES)  TRL(to[1][0][0], st_i + IMG_BYTES + 2048, t00)  TRL(to[1][0][1], st_i + IMG_BYTES + 2048, t10)  SB
            mfma_agpr<A_DV + 48>(frag(to[1][1][0], to[1][1][1]), b1);  SM_14  TRL(tq[1][0][0], st_i + 2048, t00)  TRL(tq[1][0][1], st_i + 2048, t10)  SB
            mfma_agpr<A_DK + 48>(frag(tq[1][1][0], tq[1][1][1]), e1);  SM_15  TRL(to[1][1][0], st_i + IMG_BYTES + 2048, t01)  TRL(to[1][1][1], st_i + IMG_BYTES + 2048, t11)  SB
        }
        // ---- group 3: S / dP of unit (i + 1, 0)   || softmax of unit (i, 1), scores 0..7; the last transposed fragment.  22 reads have been issued when the first
        //      wait runs: lgkmcnt(15) = the seven oldest are back -- rq0 ro0 rq1 ro1 T0 rq2, what the first five MFMAs need; then one counted wait per row fragment
        LWAIT(15, rq[0])  mfma_init<A_K + 0>(S0, rq[0], cL);   TRL(tq[1][1][0], st_i + 2048, t01)  TRL(tq[1][1][1], st_i + 2048, t11)  SM_16  SB
                          mfma_init<A_V + 0>(P0, ro[0], cD);   SM_17  SB
                          mfma_more<A_K + 4>(S0, rq[1]);       SM_18  SB
                          mfma_more<A_V + 4>(P0, ro[1]);       SM_19  SB
                          mfma_more<A_K + 8>(S0, rq[2]);       SM_20  SB
        LWAIT(14, ro[2])  mfma_more<A_V + 8>(P0, ro[2]);       SM_21  SB
        LWAIT(11, rq[3])  mfma_more<A_K + 12>(S0, rq[3]);      SM_22  SB
        LWAIT(8, ro[3])   mfma_more<A_V + 12>(P0, ro[3]);      SM_23  SB
        // ---- group 4: dV^T / dK^T of unit (i, 0)   || softmax of unit (i, 1), scores 8..15.  The last transposed read was issued eight MFMAs ago: one wait covers all
        {
            const bf16x8 b0 = pfrag(pw0, 0), e0 = pfrag(dw0, 0), b1 = pfrag(pw0, 1), e1 = pfrag(dw0, 1);
            if constexpr (!(DKDV_ABL & 512))
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(to[0][0][0]), "+v"(to[0][0][1]), "+v"(tq[0][0][0]), "+v"(tq[0][0][1]), "+v"(to[0][1][0]), "+v"(to[0][1][1]),
                         "+v"(tq[0][1][0]), "+v"(tq[0][1][1]) :: "memory");
            asm volatile("" : "+v"(to[1][0][0]), "+v"(to[1][0][1]), "+v"(tq[1][0][0]), "+v"(tq[1][0][1]), "+v"(to[1][1][0]), "+v"(to[1][1][1]),
                         "+v"(tq[1][1][0]), "+v"(tq[1][1][1]) :: "memory");
            mfma_agpr<A_DV + 0>(frag(to[0][0][0], to[0][0][1]), b0);   SM_24   SB
            mfma_agpr<A_DK + 0>(frag(tq[0][0][0], tq[0][0][1]), e0);   SM_25   SB
            mfma_agpr<A_DV + 32>(frag(to[0][1][0], to[0][1][1]), b0);  SM_26  SB
            mfma_agpr<A_DK + 32>(frag(tq[0][1][0], tq[0][1][1]), e0);  SM_27  SB
            mfma_agpr<A_DV + 0>(frag(to[1][0][0], to[1][0][1]), b1);   SM_28  SB
            mfma_agpr<A_DK + 0>(frag(tq[1][0][0], tq[1][0][1]), e1);   SM_29  SB
            mfma_agpr<A_DV + 32>(frag(to[1][1][0], to[1][1][1]), b1);  SM_30  SB
            mfma_agpr<A_DK + 32>(frag(tq[1][1][0], tq[1][1][1]), e1);  SM_31  SB
        }
        sc_n = (sc_n + SC_STEP) & (NSTATG * 1024 - 1);
    };
    // ---- the items of this workgroup (one unless PERSIST).  The ring runs on across items: slice j of the next item sits in stage (k0 + nsl + j) mod NST -----------
    int k0 = 0;                                                             // ring stage of the current item's slice 0
    for (bool first = true;; first = false) {
        bool has_next = false;
        Item nxt = cur;
        if constexpr (PERSIST) {
            has_next = item + istride < nitems;
            if (has_next) nxt = decode(item + istride);
            item_offsets(nxt, offq_n, offo_n, offs_n);                      // (the last item's trailing DMA re-reads its own first slices: harmless)
        }
        item_prologue(first, k0 * STAGE_BYTES);
        // slice i of the item runs as unrolled copy (k0 + i) mod NST, whose ring stage is a compile-time constant: the first trip starts in the middle
        for (int i = -k0; i < nsl; i += NST) {
            sfor<NST>([&](auto K) {
                constexpr int k = decltype(K)::value;
                if ((unsigned)(i + k) < (unsigned)nsl) slice_step(i + k, K);
            });
        }
        STAMP(2)
        agpr_claim();
        // the next item's K / V rows: requested before the drain and the epilogue, which hide their latency; vmcnt completes in order, so they are back before the
        // 16 stores the epilogue issues behind them
        const bf16* nbase = base;
        int nkey0 = key0;
        if constexpr (PERSIST) {
            if (has_next) {
                nbase = qkv + (int64_t)nxt.b * N * RS + nxt.h * 64;
                nkey0 = key_first + nxt.blk * (64 * NW) + wave * 64;
                load_kv(nbase, nkey0);
            }
        }
        // ---- drain: the last score of unit (nsl - 1, 1), then its dV^T / dK^T ---------------------------------------------------------------------------------------
        SM_TAIL
        SB
        {
            const bf16x8 b0 = pfrag(pw1, 0), e0 = pfrag(dw1, 0), b1 = pfrag(pw1, 1), e1 = pfrag(dw1, 1);
            mfma_agpr<A_DV + 16>(frag(to[0][0][0], to[0][0][1]), b0);  mfma_agpr<A_DK + 16>(frag(tq[0][0][0], tq[0][0][1]), e0);
            mfma_agpr<A_DV + 48>(frag(to[0][1][0], to[0][1][1]), b0);  mfma_agpr<A_DK + 48>(frag(tq[0][1][0], tq[0][1][1]), e0);
            mfma_agpr<A_DV + 16>(frag(to[1][0][0], to[1][0][1]), b1);  mfma_agpr<A_DK + 16>(frag(tq[1][0][0], tq[1][0][1]), e1);
            mfma_agpr<A_DV + 48>(frag(to[1][1][0], to[1][1][1]), b1);  mfma_agpr<A_DK + 48>(frag(tq[1][1][0], tq[1][1][1]), e1);
        }
        if constexpr (QS > 1) {
            // ---- query split: the four waves hold partial dK^T / dV^T of the SAME 64 keys.  Each drops its fp32 tile into LDS ([64 keys][dK 64 | dV 64] floats, 512 B a
            // row, 16-byte chunk c at c ^ (row & 15) inside each 256-byte half: conflict-free for these writes and for the reads below), then wave w adds the four tiles
            // of keys 16 w .. + 16 in wave order -- deterministic -- and stores whole rows
            asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // every wave is done with its rings
            char* part = smem_all + wq * 32768;
            sfor<16>([&](auto I) {
                constexpr int db = (decltype(I)::value >> 3) & 1, kb = (decltype(I)::value >> 2) & 1, jj = decltype(I)::value & 3;
                constexpr int ra = (db * 2 + kb) * 16 + 4 * jj;
                const int row = 32 * kb + r32;
                const int off = row * 512 + (((8 * db + 2 * jj + hi) ^ (row & 15)) << 4);      // floats d = 32 db + 8 jj + 4 hi .. + 4
                *reinterpret_cast<f32x4*>(part + off) = f32x4{agpr_read1<A_DK + ra>(), agpr_read1<A_DK + ra + 1>(), agpr_read1<A_DK + ra + 2>(), agpr_read1<A_DK + ra + 3>()};
                *reinterpret_cast<f32x4*>(part + off + 256) = f32x4{agpr_read1<A_DV + ra>(), agpr_read1<A_DV + ra + 1>(), agpr_read1<A_DV + ra + 2>(), agpr_read1<A_DV + ra + 3>()};
            });
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int pass = 0; pass < 8 / QS; ++pass) {
                const int row = (64 / QS) * wq + 8 * pass + (lane >> 3), c8 = lane & 7, key = key0 + row;
                f32x4 sk[2], sv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = row * 512 + (((2 * c8 + e) ^ (row & 15)) << 4);
                    sk[e] = *reinterpret_cast<const f32x4*>(smem_all + off);
                    sv[e] = *reinterpret_cast<const f32x4*>(smem_all + off + 256);
#pragma unroll
                    for (int w = 1; w < QS; ++w) {
                        sk[e] += *reinterpret_cast<const f32x4*>(smem_all + w * 32768 + off);
                        sv[e] += *reinterpret_cast<const f32x4*>(smem_all + w * 32768 + off + 256);
                    }
                    sk[e] *= (xcd & 2) ? LN2_1W : scale;
                }
                if (key < N) {
                    bf16* dst = dqkv + ((int64_t)b * N + key) * RS + h * 64 + 8 * c8;
                    *reinterpret_cast<u32x4*>(dst + D) = u32x4{cvt_pk_bf16(sk[0][0], sk[0][1]), cvt_pk_bf16(sk[0][2], sk[0][3]), cvt_pk_bf16(sk[1][0], sk[1][1]), cvt_pk_bf16(sk[1][2], sk[1][3])};
                    *reinterpret_cast<u32x4*>(dst + 2 * D) = u32x4{cvt_pk_bf16(sv[0][0], sv[0][1]), cvt_pk_bf16(sv[0][2], sv[0][3]), cvt_pk_bf16(sv[1][0], sv[1][1]), cvt_pk_bf16(sv[1][2], sv[1][3])};
                }
            }
            break;
        }
        if constexpr (PERSIST) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                   // the last MFMAs' results are readable (asm MFMAs: nobody pads this)
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // ... and the trailing re-reads have landed
            if constexpr (NW > 1) __builtin_amdgcn_s_barrier();                  // every wave is past its last read of the rings: they become the epilogue's tiles
        }

        // ---- epilogue: the wave's [64 keys][dK 64 | dV 64] tile through LDS (its own 16 KiB: row = key, 256 B, 16-byte chunk c at (c ^ (row & 7)) inside each
        // 128-byte half), then stores of eight whole rows per instruction.  Lane (hi, key r32 of block kb) holds d = 32 db + 8 jj + 4 hi .. + 4 of its key's rows ------
        if constexpr (!(DKDV_ABL & 2048)) {
            char* tile = smem + EPI_OFF + wave * 16384;
            const float dksc = (xcd & 2) ? LN2_1W : scale;      // dK = scale dS^T q = ln 2 dS^T q'
            sfor<16>([&](auto I) {
                constexpr int db = (decltype(I)::value >> 3) & 1, kb = (decltype(I)::value >> 2) & 1, jj = decltype(I)::value & 3;
                constexpr int ra = (db * 2 + kb) * 16 + 4 * jj;
                const int row = 32 * kb + r32;
                const int c16 = (4 * db + jj) ^ (row & 7), off = row * 256 + (c16 << 4) + 8 * hi;     // this lane's 8 bytes of 16-byte chunk 4 db + jj (d = 32 db + 8 jj + 4 hi)
                const f32x4 dk = f32x4{agpr_read1<A_DK + ra>(), agpr_read1<A_DK + ra + 1>(), agpr_read1<A_DK + ra + 2>(), agpr_read1<A_DK + ra + 3>()} * dksc;
                const f32x4 dv = {agpr_read1<A_DV + ra>(), agpr_read1<A_DV + ra + 1>(), agpr_read1<A_DV + ra + 2>(), agpr_read1<A_DV + ra + 3>()};
                *reinterpret_cast<u32x2*>(tile + off) = u32x2{cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3])};
                *reinterpret_cast<u32x2*>(tile + off + 128) = u32x2{cvt_pk_bf16(dv[0], dv[1]), cvt_pk_bf16(dv[2], dv[3])};
            });
            STAMP(7)
            // (each wave reads back only what it wrote itself: no barrier, the compiler's lgkmcnt wait orders the reads behind the writes)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 8 * it + (lane >> 3), c = lane & 7, key = key0 + row;
                const u32x4 vk = *reinterpret_cast<const u32x4*>(tile + row * 256 + ((c ^ (row & 7)) << 4));
                const u32x4 vv = *reinterpret_cast<const u32x4*>(tile + row * 256 + 128 + ((c ^ (row & 7)) << 4));
                if (PERSIST || key < N) {       // (PERSIST: whole blocks only -- and an unconditional store is one the compiler can count: behind conditional ones it waits for them all before the next K / V)
                    bf16* dst = dqkv + ((int64_t)b * N + key) * RS + h * 64 + 8 * c;
                    *reinterpret_cast<u32x4*>(dst + D) = vk;
                    *reinterpret_cast<u32x4*>(dst + 2 * D) = vv;
                }
            }
        }
        STAMP(3)
        agpr_claim();
        if (!PERSIST || !has_next) break;
        // ---- switch: the next item's K / V into the AGPRs (its Q / dO slices 0 .. NST - 2 are in the ring or on their way), accumulators = 0 --------------------------
        fill_kv();
        item += istride; cur = nxt; b = nxt.b; h = nxt.h; base = nbase; key0 = nkey0;
        offq = offq_n; offo = offo_n; offs = offs_n;
        gb = (gb + ngrp) & (NSTATG - 1);
        k0 = (k0 + nsl) % NST;
        STAMP_NEXT
    }
    if constexpr (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last item's trailing re-reads)
}

}  // namespace

extern "C" int devias_debug_dkdv_stamps(uint64_t* out, int32_t n) {
#ifdef DKDV_STAMP
    DEVIAS_REQUIRE(out && n > 0 && n <= 4096, "devias_debug_dkdv_stamps: bad args");
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dkdv_stamp), sizeof(unsigned long long) * 8 * n, 0, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return devias_set_error(DEVIAS_ELAUNCH, "devias_debug_dkdv_stamps: copy failed");
    }
    return DEVIAS_OK;
#else
    (void)out; (void)n;
    return devias_set_error(DEVIAS_EUNSUPPORTED, "devias_debug_dkdv_stamps: the library was not built with -DDKDV_STAMP (tools/build_variant_file.sh stamp attn_bwd1w -DDKDV_STAMP)");
#endif
}

// launched by attention.hip (mhsa_bwd_impl) behind the dQ kernel, which has written `stat`: the whole 256-key blocks of every head, then the ragged rest.
// persistent != 0 (and the XCD order, and tensors below 2 GiB so that an item is a 32-bit buffer offset): one workgroup per CU walks the 256-key blocks
int devias_attn_dkdv1w_launch(const void* qkv, const void* d_o, const float* stat, void* dqkv, int B, int N, int Npad, int H, float scale, int xcd_flag, int persistent,
                              hipStream_t st) {
    const int nfull = N / 256, rest = N - nfull * 256;
#define DKDV_ARGS (const bf16*)qkv, (const bf16*)d_o, stat, (bf16*)dqkv, N, Npad, H, B, scale, xcd_flag
    if (nfull > 0) {
        const int64_t qbytes = (int64_t)B * N * 3 * H * 64 * 2;
        const int items = nfull * H * B, ncu = devias_device_cus() & ~7;
        if (persistent && (xcd_flag & 1) && qbytes < 0x7fffffff && ncu >= 8 && items > ncu) {
            hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<4, 8, true>), dim3(ncu), dim3(256), 0, st, DKDV_ARGS, 0, nfull);
            devias_count(DEVIAS_CNT_DKDV1W_PERS);
        } else {
            const dim3 grid = (xcd_flag & 1) ? dim3(items) : dim3(nfull, H, B);
            hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<4, 8, false>), grid, dim3(256), 0, st, DKDV_ARGS, 0, nfull);
            devias_count(DEVIAS_CNT_DKDV1W);
        }
    }
    if (rest > 0) {
        devias_count(DEVIAS_CNT_DKDV1W_REST);
        const dim3 grid = (xcd_flag & 1) ? dim3(H * B) : dim3(1, H, B);
        if (rest <= 64) {
            // one wave tile per head: as many waves on it (each with a share of the query slices) as keep the launch inside ONE round of one wave per SIMD
            const int heads = H * B, slots = 4 * devias_device_cus();
            if (4 * heads <= slots) hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<1, 4, false, 4>), grid, dim3(256), 0, st, DKDV_ARGS, nfull * 256, 1);
            else if (2 * heads <= slots) hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<1, 4, false, 2>), grid, dim3(128), 0, st, DKDV_ARGS, nfull * 256, 1);
            else hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<1, 4, false, 1>), grid, dim3(64), 0, st, DKDV_ARGS, nfull * 256, 1);
        }
        else if (rest <= 128) hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<2, 4, false>), grid, dim3(128), 0, st, DKDV_ARGS, nfull * 256, 1);
        else hipLaunchKernelGGL((mhsa_bwd_dkdv1w_kernel<4, 8, false>), grid, dim3(256), 0, st, DKDV_ARGS, nfull * 256, 1);
    }
#undef DKDV_ARGS
    DEVIAS_CHECK_LAUNCH("devias_mhsa_bwd(dkdv, one wave per SIMD)");
    return DEVIAS_OK;
}
