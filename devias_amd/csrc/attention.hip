// Encoder multi-head self-attention core for gfx950 (head dim 64, non-causal, arbitrary sequence length N).
// Reference semantics: softmax((q * dh^-0.5) k^T) v  (model/modeling_slot.py:102-112); the N x N matrix is never
// materialised (online softmax forward; recompute-from-logsumexp backward).
//
// bf16 path (measured mode): MFMA 16x16x32, "key-major" score tiles.  The score tile is computed TRANSPOSED,
// S^T = K Q^T, so that after the MFMA each lane owns ONE query column (lane & 15) and 4 keys per 16-key tile
// (rows 4*(lane>>4)+r): the online-softmax row statistics are lane-local plus two cross-lane shuffles, and the
// probability tile is directly the B operand of the next MFMA (O^T = V^T P^T) with a fixed k-permutation
// (element j of lane-group g <-> key 16*(2s + (j>>2)) + 4g + (j&3)), matched on the A side by reading V^T with
// the transposing LDS read ds_read_b64_tr_b16.  No probability tile ever goes through LDS.
//   forward      : workgroup = 4 waves x 32 queries, K/V tiles of 64 keys double-staged through registers -> LDS
//   backward dQ  : same decomposition (query on the lane), also produces delta = rowsum(dO * O)
//   backward dKdV: workgroup = 4 waves x 32 keys (key on the lane, K/V fragments resident in registers),
//                  Q/dO tiles of 64 queries through LDS (row image + transposed-read image each)
// fp32 path (parity mode): straightforward VALU kernels (one thread per query / two threads per key), exact fp32.
//
// qkv layout is [B, N, 3, H, 64] exactly as produced by the fused QKV GEMM; o / d_o are [B, N, H*64].
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
// bit 1 of the kernels' `xcd` argument (bit 0: XCD-aware linear grid, bits 16..: B): the q third of qkv already holds q * scale * log2 e, rounded ONCE by its producer
// (DEVIAS_ATTN_Q_PRESCALED, ABI 167).  Forward, dQ and dK / dV kernels then multiply the SAME bf16 operands -- the backward's scores are the forward's, the saved lse fits
// them exactly -- where otherwise forward / dQ round q * c and the one-wave dK / dV kernel rounds k * c (~2x the dK / dV error at peaked logits)
enum { ATTN_QPRE = 2 };
constexpr float LN2 = 0.6931471805599453f;

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// ---- LDS images of a [64 rows][64 cols] bf16 tile (8 KiB each) ------------------------------------------
// row image: 128-byte rows, 16-byte chunk index XOR (row & 7)  -> conflict-free ds_read_b128 of [row][8 cols]
__device__ __forceinline__ int img_row_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// transposed-read image: 128-byte rows, 32-byte window XOR ((row >> 1) & 3) -> conflict-free ds_read_b64_tr_b16
__device__ __forceinline__ int img_tr_off(int row, int col) {
    return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + (col & 15) * 2;
}

// The 32x32x16 forward kernel reads V^T with 32 lanes on rows 4 hi .. + 4 of TWO adjacent windows (16 lanes each) per ds_read_b64_tr_b16 where the 16x16x32
// kernels read rows 0..7 of ONE window: under img_tr_off's swizzle rows r and r + 2 then meet in the same banks (window w of row r + 2 sits where window
// w ^ 1 of row r does) -- SQ_LDS_BANK_CONFLICT = half of the kernel's LDS cycles (profiles/r4_lds_conflicts.txt).  This swizzle moves rows r, r + 2 two
// windows apart and r, r + 4 one: the eight (row, window) pairs of a 32-lane group cover all 64 banks.
// ... and its K row image: a ds_read_b128 serves lanes i and i + 16 in the same cycle (the 16x16x32 kernels' fragments put the SAME row's next chunk there;
// here lane i + 16 holds row i + 16 with the same chunk, i.e. the same banks under `chunk ^ (row & 7)`): rows 16..31 of a 32-key block take the odd
// permutation.  Measured per launch at B = 32, N = 1568, H = 12: SQ_LDS_BANK_CONFLICT 3.01e7 -> 1.51e7 (V image) -> 0 (K image), SQ_LDS_IDX_ACTIVE 6.2e7 -> 3.2e7.
__device__ __forceinline__ int row_swz2(int row) { return (row & 7) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int img_row_off2(int row, int chunk) { return row * 128 + ((chunk ^ row_swz2(row)) << 4); }
__device__ __forceinline__ int tr_swz2(int row) { return (((row >> 1) & 1) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int img_tr_off2(int row, int col) {
    return row * 128 + ((((col >> 4) ^ tr_swz2(row))) << 5) + (col & 15) * 2;
}

// fragment of the row image: lane holds tile[row = base + (lane&15)][32*ks + 8*(lane>>4) .. +8]
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int base, int ks, int lane) {
    int row = base + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + img_row_off(row, ks * 4 + (lane >> 4)));
}
// the SAME fragment as frag_rows, read from the transposed-read image: its 32-byte-window swizzle also keeps ds_read_b128 conflict free
// (a 16-lane group of that instruction covers 16 rows x one window: 8 rows take its low half, 8 its high half, and (row & 1, (row >> 1) & 3)
// is distinct within each eight) -- so a tile that is consumed both row-wise and transposed needs ONE image, one pass of LDS writes
__device__ __forceinline__ bf16x8 frag_rows_tr(const char* img, int base, int ks, int lane) {
    int row = base + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + img_tr_off(row, 32 * ks + 8 * (lane >> 4)));
}
// fragment of the transposed image for the product over ROWS of the tile (k = tile row, permuted as in the header):
// lane holds tile[row = 16*(2s + (j>>2)) + 4g + (j&3)][col = cbase + (lane&15)], j = 0..7
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int cbase, int s, int lane) {
    int g = lane >> 4, c = lane & 15;
    int r0 = 32 * s + 4 * g + (c >> 2);
    int col = cbase + 4 * (c & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + img_tr_off(r0, col)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + img_tr_off(r0 + 16, col)));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
    return r;
}
// raw v_exp_f32: arguments here are <= 0 (or -inf), results below the normal range may flush to 0 -- harmless for softmax
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Block -> (block-in-head, head, batch).  Every block of one (batch, head) streams the same K/V (forward, dQ) or Q/dO (dK/dV) rows.  Workgroups go
// to the 8 XCDs round-robin by linear id, and each XCD has a private L2: with the plain (x = block-in-head, y = head, z = batch) order the ~13
// blocks of a head land on 8 different L2s and each fetches the head's rows from the Infinity Cache again.  With `xcd` the grid is linear
// and all blocks of a head get ids congruent mod 8: one XCD, one fetch.  (Speed only; any mapping is correct.)
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

// ---- attention dropout (Attention.attn_drop, model/modeling_slot.py:90,110: nn.Dropout on the softmax matrix) ---------------------------------
// The N x N matrix never exists, so neither does its mask: element (b, h, query i, key j) is kept when a 32-bit hash of (seed, b * H + h, i, j) is below
// keep * 2^32, and kept elements are scaled by 1 / keep.  The same function is evaluated by the forward and by both backward kernels (and restated in
// numpy by the oracle: oracle/ref_cpu.py attn_drop_mask).  Row sums (the softmax normaliser, logsumexp) are over the UN-dropped probabilities;
// delta = rowsum(dO * O) with the dropped-out O is still sum_j P_ij dP_ij, so the recompute-from-logsumexp backward carries over with
// dP_ij = mask_ij (dO_i . V_j) and dV = (P * mask)^T dO.
struct DropP { uint32_t thresh, s0, s1; float inv_keep; };
__device__ __forceinline__ uint32_t drop_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t drop_rowkey(const DropP& d, uint32_t bh, uint32_t i) { return drop_mix(drop_mix(d.s0 ^ (bh * 0x9E3779B1u)) + d.s1 + i * 0x85EBCA6Bu); }
// 0 or 1 / keep
__device__ __forceinline__ float drop_scale(const DropP& d, uint32_t rowkey, uint32_t j) { return drop_mix(rowkey + j * 0xC2B2AE35u) < d.thresh ? d.inv_keep : 0.f; }

// stage a [64][64] bf16 tile: 512 16-byte chunks, 512/NT per thread.  rows >= nrows are zero-filled.
template <int NT>
struct TileRegs {
    enum { NCH = 512 / NT };
    u32x4 v[NCH];
    __device__ __forceinline__ void load(const bf16* __restrict__ base, int64_t row_stride, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NT, row = c >> 3, ch = c & 7;
            u32x4 z = {0u, 0u, 0u, 0u};
            v[i] = (row0 + row < nrows) ? *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + row) * row_stride + ch * 8) : z;
        }
    }
    __device__ __forceinline__ void store_rows(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NT, row = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4*>(img + img_row_off(row, ch)) = v[i];
        }
    }
    __device__ __forceinline__ void store_tr(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NT, row = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4*>(img + img_tr_off(row, ch * 8)) = v[i];
        }
    }
};

// LDS-DMA staging of a [64 rows][64 cols] bf16 tile (8 x 1 KiB global_load_lds, shared by NW waves) straight into the row image
// (TR = false) or the transposed-read image (TR = true); the images' XOR swizzles are applied on the per-lane SOURCE address.
// Rows past `nrows` re-read row nrows-1 (finite data; their scores are masked / their probabilities are zero).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
template <int NW, bool TR>
__device__ __forceinline__ void dma_tile(const bf16* __restrict__ base, int64_t row_stride, int row0, int nrows, char* img, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int r8 = (wave * (8 / NW) + i) * 8;
        const int row = r8 + (lane >> 3), slot = lane & 7;
        const int chunk = TR ? ((((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1)) : (slot ^ (row & 7));
        const int grow = min(row0 + row, nrows - 1);
        const bf16* src = base + (int64_t)grow * row_stride + chunk * 8;
        __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(img + r8 * 128), 16, 0, 0);
    }
}

// The same staging through a buffer descriptor (buffer_load ... lds): descriptor + SCALAR tile offset + a loop-invariant per-lane offset, i.e. no
// vector arithmetic per tile (the per-lane 64-bit source pointers of dma_tile cost 12-15 vector instructions per wave-instruction, and these
// kernels are bound by their vector instruction count: rocprofv3 PMC, profiles/r3_attn_pmc.txt).  Rows past `nrows` are not clamped: they belong
// to the next batch entry (finite) or lie beyond the tensor (read as zero); every consumer zeroes their probabilities.
// TR: 0 = row image, 1 = transposed-read image, 2 = transposed-read image with the window swizzle of img_tr_off2 (forward kernel's V),
// 3 = row image with the chunk swizzle of img_row_off2 (forward kernel's K)
template <int NW, int TR>
struct TileDma {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t vo[8 / NW];
    int stride_bytes;
    __device__ __forceinline__ void init(const bf16* p, int64_t row_stride, int64_t elems_left, int wave, int lane) {
        const int64_t bytes = elems_left * 2;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
        stride_bytes = (int)row_stride * 2;
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
            const int row = (wave * (8 / NW) + i) * 8 + (lane >> 3), slot = lane & 7;
            const int wswz = TR == 2 ? tr_swz2(row) : ((row >> 1) & 3);
            const int chunk = TR == 0 ? (slot ^ (row & 7)) : TR == 3 ? (slot ^ row_swz2(row)) : ((((slot >> 1) ^ wswz) << 1) | (slot & 1));
            vo[i] = (uint32_t)((row * (int)row_stride + chunk * 8) * 2);
        }
    }
    __device__ __forceinline__ void issue(int row0, char* img, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass does not know the builtin)
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
            const int r8 = (wave * (8 / NW) + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(img + r8 * 128), 16, vo[i], row0 * stride_bytes, 0, 0);
        }
#else
        (void)row0; (void)img; (void)wave;
#endif
    }
};

// ======================================= forward (bf16) ===================================================
template <int QT, int NW, bool DMA, int OCC = 1>
__global__ __launch_bounds__(NW * 64, OCC) void mhsa_fwd_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                            float* __restrict__ lse, int N, int H, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) char smem[DMA ? 32768 : 16384];   // DMA: two stages of (K row image | V transposed-read image)
    char* imgK = smem;            // row image of K tile  [key][d]
    char* imgV = smem + 8192;     // transposed-read image of V tile [key][d]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadMap hm = head_map((N + NW * 16 * QT - 1) / (NW * 16 * QT), H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const int q0 = hm.blk * (NW * 16 * QT) + wave * (16 * QT);
    const float sl2 = (xcd & ATTN_QPRE) ? 1.0f : scale * LOG2E;      // (Q already holds q * scale * log2 e: DEVIAS_ATTN_Q_PRESCALED)

    bf16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = min(q0 + 16 * qt + c, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * RS + 32 * ks + 8 * g);
    }
    f32x4 acc_o[4][QT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j) acc_o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[QT], lrow[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) { mrow[j] = -INFINITY; lrow[j] = 0.f; }

    const int nkv = (N + 63) / 64;
    TileRegs<NW * 64> rk, rv;
    TileDma<NW, 0> dK;
    TileDma<NW, 1> dV;
    if constexpr (DMA) {
        const int64_t left = ((int64_t)(xcd >> 16) - b) * N * RS - h * 64;      // elements from `base` to the end of qkv
        dK.init(base + D, RS, left - D, wave, lane);
        dV.init(base + 2 * D, RS, left - 2 * D, wave, lane);
        dK.issue(0, smem, wave);
        dV.issue(0, smem + 8192, wave);
    } else {
        rk.load(base + D, RS, 0, N, tid);
        rv.load(base + 2 * D, RS, 0, N, tid);
    }
    for (int t = 0; t < nkv; ++t) {
        if constexpr (DMA) {
            // one barrier per tile: tile t has landed for every wave, and everyone is done reading tile t-1's stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            imgK = smem + (t & 1) * 16384;
            imgV = imgK + 8192;
            if (t + 1 < nkv) {
                char* nxt = smem + ((t + 1) & 1) * 16384;
                dK.issue((t + 1) * 64, nxt, wave);
                dV.issue((t + 1) * 64, nxt + 8192, wave);
            }
        } else {
            rk.store_rows(imgK, tid);
            rv.store_tr(imgV, tid);
            __syncthreads();
            if (t + 1 < nkv) {
                rk.load(base + D, RS, (t + 1) * 64, N, tid);
                rv.load(base + 2 * D, RS, (t + 1) * 64, N, tid);
            }
        }
        // a wave whose 16*QT queries all lie beyond N (the last workgroup of a head at N = 1568: 3 of its 4 waves) has staged its share of the
        // tile and takes part in the barrier; it leaves the matrix cores to the co-resident workgroups
        if constexpr (DMA) { if (q0 >= N) continue; }
        // S^T tile: acc_s[kt][qt] holds keys 16kt + 4g + r (rows) x query c (col)
        f32x4 acc_s[4][QT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < QT; ++j) acc_s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bf16x8 kf = frag_rows(imgK, 16 * kt, ks, lane);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc_s[kt][qt] = mfma(kf, qf[qt][ks], acc_s[kt][qt]);
            }
        const int k0 = t * 64;
        if (k0 + 64 > N) {                    // ragged last tile only (wave-uniform branch): mask keys >= N
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + 16 * kt + 4 * g + r >= N) {
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc_s[kt][qt][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // running max is kept in RAW score units; scale*log2(e) is folded into the exponent's FMA
            // chains of max(max(m, a), b): each pair folds into one v_max3_f32
            float mx = fmaxf(fmaxf(acc_s[0][qt][0], acc_s[0][qt][1]), acc_s[0][qt][2]);
            mx = fmaxf(fmaxf(mx, acc_s[0][qt][3]), acc_s[1][qt][0]);
            mx = fmaxf(fmaxf(mx, acc_s[1][qt][1]), acc_s[1][qt][2]);
            float mx2 = fmaxf(fmaxf(acc_s[1][qt][3], acc_s[2][qt][0]), acc_s[2][qt][1]);
            mx2 = fmaxf(fmaxf(mx2, acc_s[2][qt][2]), acc_s[2][qt][3]);
            mx2 = fmaxf(fmaxf(mx2, acc_s[3][qt][0]), acc_s[3][qt][1]);
            mx = fmaxf(fmaxf(mx, acc_s[3][qt][2]), acc_s[3][qt][3]);
            mx = fmaxf(mx, mx2);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (__any(mx > mrow[qt])) {       // some row's max grew: rescale (rare after the first tiles)
                const float mnew = fmaxf(mrow[qt], mx);
                const float alpha = fast_exp2((mrow[qt] - mnew) * sl2);
                mrow[qt] = mnew;
                lrow[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc_o[dt][qt] *= alpha;
            }
            const float nb = -mrow[qt] * sl2;
            const f32x2 sl2v = {sl2, sl2}, nbv = {nb, nb};
            f32x2 ps2 = {0.f, 0.f};                       // packed fp32: one v_pk_fma_f32 / v_pk_add_f32 per two scores
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 e = f32x2{acc_s[kt][qt][r], acc_s[kt][qt][r + 1]} * sl2v + nbv;
                    const f32x2 p = {fast_exp2(e[0]), fast_exp2(e[1])};
                    acc_s[kt][qt][r] = p[0]; acc_s[kt][qt][r + 1] = p[1];
                    ps2 += p;
                }
            lrow[qt] += ps2[0] + ps2[1];
        }
        // O^T += V^T P^T
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 pf[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) pf[qt] = pack8(acc_s[2 * s][qt], acc_s[2 * s + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8 vf = frag_tr(imgV, 16 * dt, s, lane);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc_o[dt][qt] = mfma(vf, pf[qt], acc_o[dt][qt]);
            }
        }
        if constexpr (!DMA) __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrow[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int q = q0 + 16 * qt + c;
        if (q < N) {
            const float inv = 1.0f / l;
            bf16* orow = o + ((int64_t)b * N + q) * D + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt, acc_o[dt][qt] * inv);
            if (g == 0) lse[((int64_t)b * H + h) * N + q] = (mrow[qt] * sl2 + log2f(l)) * LN2;
        }
    }
}

// ======================================= forward (bf16), 32x32x16 MFMAs ====================================
// The same decomposition as mhsa_fwd_bf16_kernel (workgroup = NW waves x 32 queries, 64-key tiles by LDS-DMA into a two-stage ring, S^T = K Q^T so
// that a query is a lane COLUMN and the probabilities are the next MFMA's operand without leaving the registers), built for the vector-issue
// budget instead of the matrix one: these kernels are bound by how many vector instructions fit beside the MFMAs (rocprofv3 PMC: 226 vector
// instructions per 32 MFMAs in the 16x16x32 kernel, matrix pipe 40 % busy), and a 32x32x16 MFMA leaves 24 of its 32 cycles to the vector
// ALU where a 16x16x32 leaves 8 of 16.  Per score the kernel issues one v_exp_f32, half a v_max3, half a v_cvt_pk and one add:
//   * Q is pre-multiplied by scale*log2(e) once (registers), and the S^T accumulators START from -m (the running row maximum, lane-constant), so
//     the MFMA delivers S' = s - m and p = exp2(S') needs no subtraction;
//   * the running maximum moves only when a tile's maximum exceeds it by more than THR = 6 (p <= 64: exact in the fp32 sums, bf16 P keeps its
//     relative precision); then -- and at the first tile -- O, l, -m and the pending S' are rescaled together, BEFORE the tile's P is formed
//     (MI355X guide T13: exponentiate a tile's P only after the decision that covers it);
//   * register pairs (r, r + 1) of the S^T tile convert straight to the B operand of O^T += V^T P^T (k order inside a 16-key step: element j of
//     lane half h is key 8 (j >> 2) + 4 h + (j & 3); the V^T fragments are read with that order by ds_read_b64_tr_b16).
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ u32x2 lds_read_tr_asm(const char* p) {
    u32x2 r;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const bf16x2 t = {(bf16)a, (bf16)b};
    return *reinterpret_cast<const unsigned*>(&t);
}

template <int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64, 2) void mhsa_fwd32_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse,
                                                                    int N, int H, float scale, int xcd, DropP drop = DropP{}) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];      // two stages of (K row image | V transposed-read image)
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, r32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadMap hm = head_map((N + NW * 32 - 1) / (NW * 32), H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const int q0 = hm.blk * (NW * 32) + wave * 32;
    const bool active = q0 < N;                                        // (a wave without a valid query only stages and synchronises)
    const float sl2 = (xcd & ATTN_QPRE) ? 1.0f : scale * LOG2E;      // (Q already holds q * scale * log2 e: DEVIAS_ATTN_Q_PRESCALED)
    constexpr float THR = 6.0f;
    uint32_t rowkey = 0;                                               // DROP: this lane's query row of the mask
    if constexpr (DROP) rowkey = drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)min(q0 + r32, N - 1));

    // Q^T as the B operand: lane (hi, query q0 + r32) holds d = 16 ks + 8 hi .. + 8, pre-scaled
    bf16x8 qf[4];
    {
        const int q = min(q0 + r32, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * RS + 16 * ks + 8 * hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16)((float)v[j] * sl2);
        }
    }
    f32x16 ot[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; negm[r] = 0.f; }
    float mrow = 0.f, lsum = 0.f;                                       // running maximum (log2 units) and this lane's share of the row sum

    // loop-invariant LDS byte offsets: K row fragments (per 16-deep step ks; second key block = + 4096), V^T fragments (per 32-d block, + 2048 per 16 keys)
    int ko[4], vo_[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ko[ks] = img_row_off2(r32, 2 * ks + hi);
    {
        const int dsub = (lane >> 4) & 1, c = lane & 15;
#pragma unroll
        for (int db = 0; db < 2; ++db) vo_[db] = img_tr_off2(4 * hi + (c >> 2), 32 * db + 16 * dsub + 4 * (c & 3));
    }

    const int nkv = (N + 63) / 64;
    TileDma<NW, 3> dK;
    TileDma<NW, 2> dV;
    {
        const int64_t left = ((int64_t)(xcd >> 16) - b) * N * RS - h * 64;
        dK.init(base + D, RS, left - D, wave, lane);
        dV.init(base + 2 * D, RS, left - 2 * D, wave, lane);
    }
    auto stage = [&](int t) -> char* { return smem + (t & 1) * 16384; };
    auto dma = [&](int t) { dK.issue(t * 64, stage(t), wave); dV.issue(t * 64, stage(t) + 8192, wave); };

    // S'^T = K Q'^T - m of tile t: two 32-key blocks, rows = keys (r & 3) + 8 (r >> 2) + 4 hi, column = query r32.  MASK: ragged last tile
    auto s_tile = [&](int t, f32x16 (&st)[2], bool mask) {
        const char* imgK = stage(t);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = mfma32(*reinterpret_cast<const bf16x8*>(imgK + kb * 4096 + ko[0]), qf[0], negm);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) st[kb] = mfma32(*reinterpret_cast<const bf16x8*>(imgK + kb * 4096 + ko[ks]), qf[ks], st[kb]);
        }
        if (mask) {                                                     // keys >= N contribute p = 0
            const int k0 = t * 64;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) st[kb][r] = -INFINITY;
        }
    };
    // maximum of a tile's S' (relative to the running maximum) over this lane's 32 keys and its partner half's; then, if some row's maximum moved by
    // more than THR (or at the first tile), O, l, -m and the pending S' take the same shift -- before the tile's P exists
    auto decide = [&](f32x16 (&st)[2], bool first) {
        float mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[0][r]), st[0][r + 1]);
        mx = fmaxf(mx, st[0][15]);
        float mx2 = fmaxf(fmaxf(st[1][0], st[1][1]), st[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx2 = fmaxf(fmaxf(mx2, st[1][r]), st[1][r + 1]);
        mx = fmaxf(mx, fmaxf(mx2, st[1][15]));
        {   // the other half's maximum by v_permlane32_swap (one vector instruction) instead of a ds_bpermute round trip through the LDS crossbar on every tile's critical path
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));      // (forward kernel 316.6 -> 309.7 us)
        }
        if (first || __any(mx > THR)) {
            const float shift = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 0.f : fast_exp2(-shift);
            mrow += shift;
            lsum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ot[0][r] *= alpha; ot[1][r] *= alpha;
                st[0][r] -= shift; st[1][r] -= shift;
                negm[r] = -mrow;
            }
        }
    };
    // P = exp2(S'), this lane's share of the row sums, and the B operands of O^T += V^T P^T (register pairs (r, r + 1) -> one packed word)
    auto soft = [&](const f32x16 (&st)[2], bf16x8 (&pf)[2][2], int t) {
        float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float p0 = fast_exp2(st[kb][8 * s2 + j]), p1 = fast_exp2(st[kb][8 * s2 + j + 1]);
                    ls0 += p0; ls1 += p1;
                    if constexpr (DROP) {                               // the sums above are of the un-dropped probabilities; O takes the dropped ones
                        const int r = 8 * s2 + j, key = t * 64 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi;      // (r + 1 is key + 1)
                        w[j >> 1] = cvt_pk_bf16(p0 * drop_scale(drop, rowkey, (uint32_t)key), p1 * drop_scale(drop, rowkey, (uint32_t)key + 1));
                    } else
                    w[j >> 1] = cvt_pk_bf16(p0, p1);
                }
                pf[kb][s2] = *reinterpret_cast<const bf16x8*>(&w);
            }
        lsum += ls0 + ls1;
    };
    // O^T += V^T P^T of tile t: 16-key steps kk = 2 kb + s2; transposed reads from inline assembly (before the builtin the compiler waits
    // vmcnt(0), i.e. for the LDS-DMA in flight -- gemm.hip)
    auto pv_tile = [&](int t, const bf16x8 (&pf)[2][2]) {
        const char* imgV = stage(t) + 8192;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x2 vlo[4], vhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = 2 * half + (i >> 1), db = i & 1;
                const char* pv = imgV + kk * 2048 + vo_[db];
                vlo[i] = lds_read_tr_asm(pv);
                vhi[i] = lds_read_tr_asm(pv + 1024);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]), "+v"(vlo[2]), "+v"(vhi[2]), "+v"(vlo[3]), "+v"(vhi[3]) :: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = 2 * half + (i >> 1), db = i & 1;
                const u32x4 w = {vlo[i][0], vlo[i][1], vhi[i][0], vhi[i][1]};
                ot[db] = mfma32(*reinterpret_cast<const bf16x8*>(&w), pf[kk >> 1][kk & 1], ot[db]);
            }
        }
    };

    // One tile per iteration and barrier: S' (8 MFMAs) -> decision -> softmax arithmetic -> O^T += V^T P^T (8 MFMAs); the matrix and the vector phases of
    // a wave do not overlap each other -- the 3-4 waves per SIMD (127 registers) do.  A software-pipelined form (S' of tile t + 1 beside the softmax
    // arithmetic of tile t, three LDS stages) was built and measured SLOWER: 355 vs 305 us -- it needs 238 registers (two waves per SIMD), and forced
    // to 168 it spills (2.7 ms).
    dma(0);
    for (int t = 0; t < nkv; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // tile t has landed for this wave ...
        __syncthreads();                                                // ... and for every wave; everyone is done reading tile t - 1's stage
        if (t + 1 < nkv) dma(t + 1);
        if (!active) continue;
        f32x16 st[2];
        bf16x8 pf[2][2];
        s_tile(t, st, t + 1 == nkv && (N & 63) != 0);
        decide(st, t == 0);
        soft(st, pf, t);
        pv_tile(t, pf);
    }
    if (q0 + r32 < N) {
        float l = lsum + __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / l;
        const int q = q0 + r32;
        bf16* orow = o + ((int64_t)b * N + q) * D + h * 64 + 4 * hi;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                store4(orow + 32 * db + 8 * rq, f32x4{ot[db][4 * rq], ot[db][4 * rq + 1], ot[db][4 * rq + 2], ot[db][4 * rq + 3]} * inv);
        if (hi == 0) lse[((int64_t)b * H + h) * N + q] = (mrow + log2f(l)) * LN2;
    } else if (active) {
        (void)__shfl_xor(lsum, 32, 64);
    }
}

// Column sums of a gradient tile for the bias gradient (q_bias / v_bias: the sum of dQ / dV over all rows), taken from the fp32 accumulators BEFORE they are
// rounded and stored: lane (g, c) holds, per 16-wide d-tile dt, the 4 values d = 16 dt + 4 g + r of row c of each of its NT row tiles.  Sum over the valid row
// tiles, over the 16 lanes c of a group (xor 1, 2, 4, 8), over the 4 waves through LDS in wave order: one [64] partial per workgroup, written to
// part[(b * nblk + blk) * D + h * 64 + d] and reduced over (b, blk) in a fixed order by the caller's second stage (devias_colsum_finish): deterministic, no atomics.
// Replaces two column-sum passes over the stored tensor per encoder block (77 MB read each at ViT-B).  `smem` is free: the caller has synchronised after its last tile.
template <int NT>
__device__ __forceinline__ void bias_partials(const f32x4 (&acc)[4][NT], const bool (&valid)[NT], float mul, char* smem, float* __restrict__ part, int64_t slot,
                                              int tid, int lane, int wave) {
    // every lane drops its 16 row-tile sums into LDS ([wave][d = 16 dt + 4 g + r][c], 16 KB), then thread d < 64 of the workgroup adds the 4 x 16 entries of its
    // column in a fixed order (cross-lane shuffles cost this epilogue more than the 64 LDS reads do)
    float* red = reinterpret_cast<float*>(smem);
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (valid[t]) v += acc[dt][t] * mul;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 64 + 16 * dt + 4 * g + r) * 17 + c] = v[r];      // (17: the 16 lanes of a group write 16 different banks, the 4 groups too)
    }
    __syncthreads();
    if (tid < 64) {
        float t4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* p = red + (w * 64 + tid) * 17;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) a += p[i];
            t4[w] = a;
        }
        part[slot + tid] = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
}

// ======================================= backward dQ (bf16) ==============================================
template <int QT, int NW, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void mhsa_bwd_dq_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                               const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                               int N, int H, float scale, int xcd, DropP drop = DropP{}, float* __restrict__ part_q = nullptr,
                                                               float* __restrict__ stat = nullptr, int Npad = 0) {
    __shared__ __attribute__((aligned(16))) char smem[32768];     // two stages of (K image | V image), filled by LDS-DMA one tile ahead
    // per stage: K, one image for both uses -- row reads (S^T = K Q^T) and transposed reads (dQ^T = K^T dS^T) -- | V rows (dP^T = V dO^T)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const HeadMap hm = head_map((N + NW * 16 * QT - 1) / (NW * 16 * QT), H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const int q0 = hm.blk * (NW * 16 * QT) + wave * (16 * QT);
    const float sl2 = (xcd & ATTN_QPRE) ? 1.0f : scale * LOG2E;      // (Q already holds q * scale * log2 e: DEVIAS_ATTN_Q_PRESCALED)

    bf16x8 qf[QT][2], dof[QT][2];
    float lse2[QT], dl[QT];
    uint32_t rowkey[QT];                                               // DROP: the mask rows of this lane's queries
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = min(q0 + 16 * qt + c, N - 1);
        rowkey[qt] = DROP ? drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)q) : 0u;
        const bf16* orow = o + ((int64_t)b * N + q) * D + h * 64;
        const bf16* dorow = d_o + ((int64_t)b * N + q) * D + h * 64;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qt][ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * RS + 32 * ks + 8 * g);
            dof[qt][ks] = *reinterpret_cast<const bf16x8*>(dorow + 32 * ks + 8 * g);
            bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + 32 * ks + 8 * g);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)dof[qt][ks][j] * (float)of[j];
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        dl[qt] = part;
        lse2[qt] = lse[((int64_t)b * H + h) * N + q] * LOG2E;
        if constexpr (!DROP) {                                        // Q pre-multiplied by scale * log2 e (as the forward kernel does): with the row constants below no per-score arithmetic is left before the v_exp
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (bf16)((float)qf[qt][ks][j] * sl2);
        }
        if (g == 0 && q0 + 16 * qt + c < N) delta[((int64_t)b * H + h) * N + q] = part;
        // the row statistics in the form the one-wave-per-SIMD dK / dV kernel (attn_bwd1w.hip) streams them by LDS-DMA: [B, H, Npad / 32, 2, 32], per 32-query
        // slice -lse * log2 e | -delta, the padding rows N .. Npad - 1 as -inf | 0 (their probabilities are then exactly zero without masking code)
        if (stat && g == 0 && q0 + 16 * qt + c < Npad) {
            const int qq = q0 + 16 * qt + c;
            const bool in = qq < N;
            float* sp = stat + ((int64_t)b * H + h) * 2 * Npad + (qq >> 5) * 64 + (qq & 31);
            sp[0] = in ? -lse2[qt] : -INFINITY;
            sp[32] = in ? -part : 0.f;
        }
    }
    f32x4 acc_dq[4][QT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j) acc_dq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cl4[QT], cd4[QT];                                           // the row constants as MFMA C operands (no dropout)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { cl4[qt] = f32x4{-lse2[qt], -lse2[qt], -lse2[qt], -lse2[qt]}; cd4[qt] = f32x4{-dl[qt], -dl[qt], -dl[qt], -dl[qt]}; }

    const int nkv = (N + 63) / 64;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    TileDma<NW, 1> dK;
    TileDma<NW, 0> dV;
    {
        const int64_t left = ((int64_t)(xcd >> 16) - b) * N * RS - h * 64;      // elements from `base` to the end of qkv
        dK.init(base + D, RS, left - D, wv, lane);
        dV.init(base + 2 * D, RS, left - 2 * D, wv, lane);
    }
    dK.issue(0, smem, wv);
    dV.issue(0, smem + 8192, wv);
    // (two tiles per trip, the LDS stage a compile-time constant in each: every fragment address is then lane offset + immediate -- the run-time stage cost ~36
    // vector instructions of address arithmetic per tile, in a kernel bound by vector issue)
    auto tile = [&](int t, auto STG) {
        // one barrier per tile: tile t has landed for every wave, and everyone is done reading tile t-1's stage (rows past N re-read row N-1:
        // finite data whose probabilities are exactly zero)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* imgKt = smem + decltype(STG)::value * 16384;
        const char* imgV = imgKt + 8192;
        if (t + 1 < nkv) {
            char* nxt = smem + (1 - decltype(STG)::value) * 16384;
            dK.issue((t + 1) * 64, nxt, wv);
            dV.issue((t + 1) * 64, nxt + 8192, wv);
        }
        if (q0 < N) {                         // (waves without a valid query only stage and synchronise: see the forward kernel)
        // The 64-key tile in two halves of 32 keys (16-key tiles kt = 2s, 2s + 1), each run to completion -- S^T and dP^T (8 QT MFMAs), the softmax arithmetic,
        // dQ^T += K^T dS^T (4 QT MFMAs) -- so that only HALF a tile of scores and dP is ever live: 32 registers fewer than the whole-tile form, which is what
        // takes the kernel from 194 to 162 registers = three waves per SIMD instead of two: -6.5 % on the kernel, -0.32 ms on the step.  Same MFMAs on the same operands in the same accumulation order:
        // bitwise equal to the whole-tile form.
        const int k0 = t * 64;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // Without dropout the row constants ride in the MFMAs' C operand (a query is a lane COLUMN of these tiles: the constant is the same in all four
            // accumulator registers): S' = s * scale * log2 e - lse2 and dP - delta come out of the matrix cores, and a score costs v_exp, v_mul and half a
            // v_cvt_pk instead of six vector instructions (these kernels are bound by vector issue, profiles/r3_attn_pmc.txt)
            f32x4 acc_s[2][QT], acc_dp[2][QT];
            if constexpr (DROP) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < QT; ++j) { acc_s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int kt = 2 * s + h2;
                    bf16x8 kf = frag_rows_tr(imgKt, 16 * kt, ks, lane);
                    bf16x8 vf = frag_rows(imgV, 16 * kt, ks, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        if (!DROP && ks == 0) {
                            acc_s[h2][qt] = mfma(kf, qf[qt][ks], cl4[qt]);
                            acc_dp[h2][qt] = mfma(vf, dof[qt][ks], cd4[qt]);
                        } else {
                            acc_s[h2][qt] = mfma(kf, qf[qt][ks], acc_s[h2][qt]);
                            acc_dp[h2][qt] = mfma(vf, dof[qt][ks], acc_dp[h2][qt]);
                        }
                    }
                }
            if (k0 + 64 > N) {                // ragged last tile only: p = exp2(-inf) = 0 for keys >= N
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * (2 * s + h2) + 4 * g + r >= N) {
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt) acc_s[h2][qt][r] = -INFINITY;
                        }
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const f32x2 sl2v = {sl2, sl2}, nl = {-lse2[qt], -lse2[qt]}, dlv = {dl[qt], dl[qt]};   // packed fp32: two scores per VALU op
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {                  // (scalar on purpose: packed fp32 instructions beside MFMAs cost more issue time than the two they replace)
                            if constexpr (!DROP) {
                                acc_s[h2][qt][r + u] = fast_exp2(acc_s[h2][qt][r + u]) * acc_dp[h2][qt][r + u];                            // p (dP - delta) = dS^T / scale
                            } else {
                            const float p1 = fast_exp2(acc_s[h2][qt][r + u] * sl2v[0] + nl[0]);
                            float dp1 = acc_dp[h2][qt][r + u];
                            dp1 *= drop_scale(drop, rowkey[qt], (uint32_t)(k0 + 16 * (2 * s + h2) + 4 * g + r + u));   // dP_ij = mask_ij (dO_i . V_j)
                            acc_s[h2][qt][r + u] = p1 * (dp1 - dlv[0]);                                                                     // dS^T / scale (scale applied once at the end)
                            }
                        }
                    }
            }
            bf16x8 dsf[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) dsf[qt] = pack8(acc_s[0][qt], acc_s[1][qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8 kf = frag_tr(imgKt, 16 * dt, s, lane);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc_dq[dt][qt] = mfma(kf, dsf[qt], acc_dq[dt][qt]);
            }
        }
        }
    };
    for (int t = 0; t < nkv; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nkv) tile(t + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + 16 * qt + c;
        if (q < N) {
            bf16* row = dqkv + ((int64_t)b * N + q) * RS + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(row + 16 * dt, acc_dq[dt][qt] * scale);
        }
    }
    if (part_q) {                                                 // (workgroup-uniform) q_bias gradient partials
        __syncthreads();                                          // every wave is done with the last tile's LDS stage
        // everything the epilogue needs is derived again from the (laundered) ids: nothing of it stays live across the key loop, whose register budget
        // (three waves per SIMD) and schedule are then those of the kernel without the epilogue
        int bid = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, t2 = threadIdx.x;
        asm volatile("" : "+s"(bid), "+s"(by), "+s"(bz), "+v"(t2));
        const int nblk = (N + NW * 16 * QT - 1) / (NW * 16 * QT);
        int blk2, h2, b2;
        if (xcd & 1) { const int x = bid & 7, slot = bid >> 3, hidx = (slot / nblk) * 8 + x; blk2 = slot - (slot / nblk) * nblk; h2 = hidx % H; b2 = hidx / H; }
        else { blk2 = bid; h2 = by; b2 = bz; }
        const int lane2 = t2 & 63, wave2 = t2 >> 6, q02 = blk2 * (NW * 16 * QT) + wave2 * (16 * QT);
        bool valid[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) valid[qt] = q02 + 16 * qt + (lane2 & 15) < N;
        bias_partials<QT>(acc_dq, valid, scale, smem, part_q, ((int64_t)b2 * nblk + blk2) * (H * 64) + h2 * 64, t2, lane2, wave2);
    }
}

// ======================================= backward dK, dV (bf16) ==========================================
// KT = 16-key tiles per wave.  2 = 32 keys per wave, 128 per workgroup, 252 registers, two waves per SIMD: the shipped form.  1 (16 / 64 keys: 114 registers, four
// waves per SIMD) was measured and is not instantiated: the step is 0.38 ms SLOWER -- every wave still streams the whole Q / dO tile through LDS for half the
// keys (profiles/r4_attn_occupancy.txt)
template <bool DROP = false, int KT = 2>
__global__ __launch_bounds__(256) void mhsa_bwd_dkdv_bf16_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 bf16* __restrict__ dqkv, int N, int H, float scale, int xcd, DropP drop = DropP{}, float* __restrict__ part_v = nullptr) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 2 * 512];   // two stages of (Q image | dO image) filled by LDS-DMA one tile ahead, + their row statistics
    char* imgQt = smem;             // Q, one image: row reads (S = Q K^T) and transposed reads (dK^T = Q^T dS)
    char* imgOt = smem + 8192;      // dO, one image: row reads (dP = dO V^T) and transposed reads (dV^T = dO^T P)
    float* s_stat = reinterpret_cast<float*>(smem + 32768);  // per stage: [64] log2-domain logsumexp (+inf for invalid rows) | [64] delta
    const float* s_lse = s_stat;
    const float* s_dl = s_stat + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const HeadMap hm = head_map((N + 64 * KT - 1) / (64 * KT), H, xcd >> 16, (xcd & 1) != 0);
    const int h = hm.h, b = hm.b;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const bf16* base = qkv + (int64_t)b * N * RS + h * 64;
    const bf16* dobase = d_o + (int64_t)b * N * D + h * 64;
    const int key0 = hm.blk * (64 * KT) + wave * (16 * KT);
    const float sl2 = (xcd & ATTN_QPRE) ? 1.0f : scale * LOG2E;      // (Q already holds q * scale * log2 e: DEVIAS_ATTN_Q_PRESCALED)

    bf16x8 kreg[KT][2], vreg[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        int key = min(key0 + 16 * kt + c, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kreg[kt][ks] = *reinterpret_cast<const bf16x8*>(base + D + (int64_t)key * RS + 32 * ks + 8 * g);
            vreg[kt][ks] = *reinterpret_cast<const bf16x8*>(base + 2 * D + (int64_t)key * RS + 32 * ks + 8 * g);
        }
    }
    f32x4 acc_dk[4][KT], acc_dv[4][KT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) { acc_dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nq = (N + 63) / 64;
    const float* lse_bh = lse + ((int64_t)b * H + h) * N;
    const float* dl_bh = delta + ((int64_t)b * H + h) * N;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // row statistics of query tile r0 / 64 for thread tid < 128 (rows past N: p = exp2(-inf) = 0 whatever the re-read Q / dO rows hold)
    auto stat_load = [&](int r0) -> float {
        if (tid < 64) return r0 + tid < N ? lse_bh[r0 + tid] * LOG2E : INFINITY;
        return r0 + tid - 64 < N ? dl_bh[r0 + tid - 64] : 0.f;
    };
    float rstat = 0.f;
    TileDma<4, 1> dQ_, dO_;
    dQ_.init(base, RS, ((int64_t)(xcd >> 16) - b) * N * RS - h * 64, wv, lane);
    dO_.init(dobase, D, ((int64_t)(xcd >> 16) - b) * N * D - h * 64, wv, lane);
    dQ_.issue(0, smem, wv);
    dO_.issue(0, smem + 8192, wv);
    if (tid < 128) {
        s_stat[tid] = stat_load(0);
        if (nq > 1) rstat = stat_load(64);      // always one tile ahead of the LDS copy
    }
    for (int t = 0; t < nq; ++t) {
        // one barrier per tile: tile t (images by LDS-DMA, statistics by the ds_write below / above) is complete for every wave, and
        // everyone is done reading tile t-1's stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        imgQt = smem + (t & 1) * 16384;
        imgOt = imgQt + 8192;
        s_lse = s_stat + (t & 1) * 128;
        s_dl = s_lse + 64;
        if (t + 1 < nq) {
            char* nxt = smem + ((t + 1) & 1) * 16384;
            dQ_.issue((t + 1) * 64, nxt, wv);
            dO_.issue((t + 1) * 64, nxt + 8192, wv);
            if (tid < 128) {
                s_stat[((t + 1) & 1) * 128 + tid] = rstat;
                if (t + 2 < nq) rstat = stat_load((t + 2) * 64);
            }
        }
        if (key0 < N) {                       // (waves without a valid key only stage and synchronise: see the forward kernel)
        // S and dP tiles: acc[qt][kt] holds queries 16qt + 4g + r (rows) x key c (col)
        f32x4 acc_s[4][KT], acc_dp[4][KT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j) { acc_s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                bf16x8 qfr = frag_rows_tr(imgQt, 16 * qt, ks, lane);
                bf16x8 dofr = frag_rows_tr(imgOt, 16 * qt, ks, lane);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    acc_s[qt][kt] = mfma(qfr, kreg[kt][ks], acc_s[qt][kt]);
                    acc_dp[qt][kt] = mfma(dofr, vreg[kt][ks], acc_dp[qt][kt]);
                }
            }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(s_lse + 16 * qt + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(s_dl + 16 * qt + 4 * g);
            const f32x2 sl2v = {sl2, sl2};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {                                    // packed fp32: two scores per VALU op
#pragma unroll
                    for (int u = 0; u < 2; ++u) {                      // (scalar on purpose: see the dQ kernel)
                        const float p1 = fast_exp2(acc_s[qt][kt][r + u] * sl2v[0] - l4[r + u]);
                        float dp1 = acc_dp[qt][kt][r + u], pd1 = p1;
                        if constexpr (DROP) {                          // query 64 t + 16 qt + 4 g + r + u, key key0 + 16 kt + c: dV takes P * mask, dP = mask (dO . V)
                            const float m1 = drop_scale(drop, drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)min(t * 64 + 16 * qt + 4 * g + r + u, N - 1)), (uint32_t)(key0 + 16 * kt + c));
                            dp1 *= m1; pd1 = p1 * m1;
                        }
                        acc_s[qt][kt][r + u] = pd1;
                        acc_dp[qt][kt][r + u] = p1 * (dp1 - d4[r + u]);                                          // dS / scale
                    }
                }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 pf[KT], dsf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                pf[kt] = pack8(acc_s[2 * s][kt], acc_s[2 * s + 1][kt]);
                dsf[kt] = pack8(acc_dp[2 * s][kt], acc_dp[2 * s + 1][kt]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8 dot = frag_tr(imgOt, 16 * dt, s, lane);
                bf16x8 qt_ = frag_tr(imgQt, 16 * dt, s, lane);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    acc_dv[dt][kt] = mfma(dot, pf[kt], acc_dv[dt][kt]);
                    acc_dk[dt][kt] = mfma(qt_, dsf[kt], acc_dk[dt][kt]);
                }
            }
        }
        }
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = key0 + 16 * kt + c;
        if (key < N) {
            bf16* row = dqkv + ((int64_t)b * N + key) * RS + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store4(row + D + 16 * dt, acc_dk[dt][kt] * ((xcd & ATTN_QPRE) ? LN2 : scale));      // (dK = scale dS^T q = ln 2 dS^T q' with q' = q scale log2 e)
                store4(row + 2 * D + 16 * dt, acc_dv[dt][kt]);
            }
        }
    }
    if (part_v) {                                                 // (workgroup-uniform) v_bias gradient partials (ids derived again: see the dQ kernel)
        __syncthreads();
        int bid = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, t2 = threadIdx.x;
        asm volatile("" : "+s"(bid), "+s"(by), "+s"(bz), "+v"(t2));
        const int nblk = (N + 64 * KT - 1) / (64 * KT);
        int blk2, h2, b2;
        if (xcd & 1) { const int x = bid & 7, slot = bid >> 3, hidx = (slot / nblk) * 8 + x; blk2 = slot - (slot / nblk) * nblk; h2 = hidx % H; b2 = hidx / H; }
        else { blk2 = bid; h2 = by; b2 = bz; }
        const int lane2 = t2 & 63, wave2 = t2 >> 6, key02 = blk2 * (64 * KT) + wave2 * (16 * KT);
        bool valid[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) valid[kt] = key02 + 16 * kt + (lane2 & 15) < N;
        bias_partials<KT>(acc_dv, valid, 1.0f, smem, part_v, ((int64_t)b2 * nblk + blk2) * (H * 64) + h2 * 64, t2, lane2, wave2);
    }
}

// ======================================= fp32 parity kernels ==============================================
// forward: 128 threads, one query per thread (q and o in registers), K/V tiles of 32 keys broadcast from LDS
template <bool DROP = false>
__global__ __launch_bounds__(128) void mhsa_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                           float* __restrict__ lse, int N, int H, float scale, DropP drop = DropP{}) {
    __shared__ __attribute__((aligned(16))) float sk[32][64];
    __shared__ __attribute__((aligned(16))) float sv[32][64];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const float* base = qkv + (int64_t)b * N * RS + h * 64;
    const int q = blockIdx.x * 128 + tid;
    const int qc = min(q, N - 1);
    float qr[64], acc[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)qc * RS + d);
        qr[d] = v[0] * scale; qr[d + 1] = v[1] * scale; qr[d + 2] = v[2] * scale; qr[d + 3] = v[3] * scale;
        acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const uint32_t rowkey = DROP ? drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)qc) : 0u;
    for (int k0 = 0; k0 < N; k0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * 16; i += 128) {
            int r = i >> 4, ch = (i & 15) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bool ok = k0 + r < N;
            *reinterpret_cast<f32x4*>(&sk[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(base + D + (int64_t)(k0 + r) * RS + ch) : z;
            *reinterpret_cast<f32x4*>(&sv[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(base + 2 * D + (int64_t)(k0 + r) * RS + ch) : z;
        }
        __syncthreads();
        const int nk = min(32, N - k0);
        for (int j = 0; j < nk; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s += qr[d] * sk[j][d];
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), p = expf(s - mn);
            m = mn;
            l = l * alpha + p;
            const float pd = DROP ? p * drop_scale(drop, rowkey, (uint32_t)(k0 + j)) : p;
#pragma unroll
            for (int d = 0; d < 64; ++d) acc[d] = acc[d] * alpha + pd * sv[j][d];
        }
    }
    if (q < N) {
        const float inv = 1.0f / l;
        float* orow = o + ((int64_t)b * N + q) * D + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            f32x4 v = {acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + d) = v;
        }
        lse[((int64_t)b * H + h) * N + q] = m + logf(l);
    }
}

// backward dQ (+ delta): one query per thread
template <bool DROP = false>
__global__ __launch_bounds__(128) void mhsa_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                              const float* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ delta, float* __restrict__ dqkv,
                                                              int N, int H, float scale, DropP drop = DropP{}) {
    __shared__ __attribute__((aligned(16))) float sk[32][64];
    __shared__ __attribute__((aligned(16))) float sv[32][64];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const float* base = qkv + (int64_t)b * N * RS + h * 64;
    const int q = blockIdx.x * 128 + tid;
    const int qc = min(q, N - 1);
    float qr[64], dor[64], dq[64];
    float dl = 0.f;
    const float* orow = o + ((int64_t)b * N + qc) * D + h * 64;
    const float* dorow = d_o + ((int64_t)b * N + qc) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) {
        qr[d] = base[(int64_t)qc * RS + d];
        dor[d] = dorow[d];
        dl += dor[d] * orow[d];
        dq[d] = 0.f;
    }
    const float ls = lse[((int64_t)b * H + h) * N + qc];
    const uint32_t rowkey = DROP ? drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)qc) : 0u;
    if (q < N) delta[((int64_t)b * H + h) * N + q] = dl;
    for (int k0 = 0; k0 < N; k0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * 16; i += 128) {
            int r = i >> 4, ch = (i & 15) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bool ok = k0 + r < N;
            *reinterpret_cast<f32x4*>(&sk[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(base + D + (int64_t)(k0 + r) * RS + ch) : z;
            *reinterpret_cast<f32x4*>(&sv[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(base + 2 * D + (int64_t)(k0 + r) * RS + ch) : z;
        }
        __syncthreads();
        const int nk = min(32, N - k0);
        for (int j = 0; j < nk; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) { s += qr[d] * sk[j][d]; dp += dor[d] * sv[j][d]; }
            const float p = expf(s * scale - ls);
            if constexpr (DROP) dp *= drop_scale(drop, rowkey, (uint32_t)(k0 + j));
            const float ds = p * (dp - dl) * scale;
#pragma unroll
            for (int d = 0; d < 64; ++d) dq[d] += ds * sk[j][d];
        }
    }
    if (q < N) {
        float* row = dqkv + ((int64_t)b * N + q) * RS + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) row[d] = dq[d];
    }
}

// backward dK/dV: two threads per key (each owns 32 of the 64 head dims); 128 threads = 64 keys per workgroup
template <bool DROP = false>
__global__ __launch_bounds__(128) void mhsa_bwd_dkdv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, int N, int H, float scale, DropP drop = DropP{}) {
    __shared__ __attribute__((aligned(16))) float sq[32][64];
    __shared__ __attribute__((aligned(16))) float sdo[32][64];
    __shared__ float sl[32], sd[32];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int64_t RS = 3 * (int64_t)D;
    const float* base = qkv + (int64_t)b * N * RS + h * 64;
    const float* dobase = d_o + (int64_t)b * N * D + h * 64;
    const int key = blockIdx.x * 64 + (tid >> 1);
    const int half = (tid & 1) * 32;
    const int kc = min(key, N - 1);
    float kr[32], vr[32], dk[32], dv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        kr[d] = base[D + (int64_t)kc * RS + half + d];
        vr[d] = base[2 * D + (int64_t)kc * RS + half + d];
        dk[d] = 0.f; dv[d] = 0.f;
    }
    for (int q0 = 0; q0 < N; q0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * 16; i += 128) {
            int r = i >> 4, ch = (i & 15) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bool ok = q0 + r < N;
            *reinterpret_cast<f32x4*>(&sq[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(base + (int64_t)(q0 + r) * RS + ch) : z;
            *reinterpret_cast<f32x4*>(&sdo[r][ch]) = ok ? *reinterpret_cast<const f32x4*>(dobase + (int64_t)(q0 + r) * D + ch) : z;
        }
        if (tid < 32) {
            bool ok = q0 + tid < N;
            sl[tid] = ok ? lse[((int64_t)b * H + h) * N + q0 + tid] : INFINITY;
            sd[tid] = ok ? delta[((int64_t)b * H + h) * N + q0 + tid] : 0.f;
        }
        __syncthreads();
        const int nq = min(32, N - q0);
        for (int i = 0; i < nq; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += sq[i][half + d] * kr[d]; dp += sdo[i][half + d] * vr[d]; }
            s += __shfl_xor(s, 1, 64);
            dp += __shfl_xor(dp, 1, 64);
            const float p = expf(s * scale - sl[i]);
            float pd = p;
            if constexpr (DROP) {
                const float mk = drop_scale(drop, drop_rowkey(drop, (uint32_t)(b * H + h), (uint32_t)(q0 + i)), (uint32_t)kc);
                dp *= mk; pd = p * mk;
            }
            const float ds = p * (dp - sd[i]) * scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) { dv[d] += pd * sdo[i][half + d]; dk[d] += ds * sq[i][half + d]; }
        }
    }
    if (key < N) {
        float* row = dqkv + ((int64_t)b * N + key) * RS + h * 64 + half;
#pragma unroll
        for (int d = 0; d < 32; ++d) { row[D + d] = dk[d]; row[2 * D + d] = dv[d]; }
    }
}

}  // namespace

// process-wide options, read from the environment once; devias_set_option("attn_cfg" | "attn_xcd", v) changes them at run time
namespace {
struct AttnKnobs { int cfg, xcd, bias_fused, dkdv, qpre; };
AttnKnobs& attn_knobs() {
    static AttnKnobs k = [] {
        AttnKnobs x;
        const char* e = getenv("DEVIAS_ATTN_CFG"); x.cfg = e ? atoi(e) : 0;
        e = getenv("DEVIAS_ATTN_XCD"); x.xcd = e ? atoi(e) : 1;
        e = getenv("DEVIAS_ATTN_BIAS_FUSED"); x.bias_fused = e ? atoi(e) : 1;
        e = getenv("DEVIAS_ATTN_DKDV"); x.dkdv = e ? atoi(e) : 1;
        e = getenv("DEVIAS_ATTN_QPRE"); x.qpre = e ? atoi(e) : 1;
        return x;
    }();
    return k;
}
}  // namespace
static int* attn_option_slot(const char* name) {
    if (!strcmp(name, "attn_cfg")) return &attn_knobs().cfg;
    if (!strcmp(name, "attn_xcd")) return &attn_knobs().xcd;
    if (!strcmp(name, "attn_qpre")) return &attn_knobs().qpre;              // 1 (default): a HOST-side policy read by devias_amd/modeling_slot.py -- bf16 encoder blocks without attention dropout run their qkv GEMM on a q-scaled weight copy and the attention kernels with DEVIAS_ATTN_Q_PRESCALED; 0: q unscaled, every kernel applies scale * log2 e itself (A/B aid).  The library itself follows the caller's flags / devias_block_args.WqkvS
    if (!strcmp(name, "attn_dkdv")) return &attn_knobs().dkdv;              // 1 (default): dK / dV by the one-wave-per-SIMD kernel (attn_bwd1w.hip), one workgroup per 256-key block; 2: the same kernel, one persistent workgroup per CU (the kernel alone -2 %, the step +-0: DESIGN.md section 5 round 5); 0: the two-waves-per-SIMD kernel
    if (!strcmp(name, "attn_bias_fused")) return &attn_knobs().bias_fused;      // 0: devias_mhsa_bwd_bias takes the bias gradients by column-sum passes in bf16 too (A/B aid)
    return nullptr;
}
int devias_attn_set_option(const char* name, int value) {
    int* slot = attn_option_slot(name);
    if (slot) *slot = value;
    return slot != nullptr;
}
int devias_attn_get_option(const char* name, int* value) {
    const int* slot = attn_option_slot(name);
    if (slot) *value = *slot;
    return slot != nullptr;
}

// bit 0: XCD-aware linear grid (needs B*H % 8 == 0); bits 16..: B.  Option attn_xcd = 0 restores the plain 3-D grid.
static int attn_xcd_flag(int B, int H) {
    return (B << 16) | ((attn_knobs().xcd && ((B * H) % 8 == 0)) ? 1 : 0);
}

// keep >= 1: no dropout.  thresh = floor(keep * 2^32) (at most 2^32 - 1), seed = (s1 << 32) | s0
static bool drop_params(float keep, uint64_t seed, DropP& d) {
    if (!(keep < 1.0f)) return false;
    const double t = floor((double)keep * 4294967296.0);
    d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.s0 = (uint32_t)seed; d.s1 = (uint32_t)(seed >> 32);
    d.inv_keep = 1.0f / keep;
    return true;
}

static int mhsa_fwd_impl(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                         int32_t dtype, float keep, uint64_t seed, void* stream, int32_t flags = 0) {
    hipStream_t st = (hipStream_t)stream;
    DropP dp{};
    const bool drop = drop_params(keep, seed, dp);
    DEVIAS_REQUIRE(qkv && o && lse && B > 0 && N > 0 && H > 0, "devias_mhsa_fwd: bad args");
    DEVIAS_REQUIRE(aligned16(qkv) && aligned16(o), "devias_mhsa_fwd: qkv/o must be 16-byte aligned");
    DEVIAS_REQUIRE(H <= 65535 && B <= 65535, "devias_mhsa_fwd: H and B must be <= 65535");
    DEVIAS_REQUIRE((flags & ~DEVIAS_ATTN_Q_PRESCALED) == 0 && (flags == 0 || dtype == DEVIAS_BF16), "devias_mhsa_fwd: bad flags %d (DEVIAS_ATTN_Q_PRESCALED: bf16 only)", flags);
    if (dtype == DEVIAS_BF16)
        {
        const int cfg = attn_knobs().cfg;
        const int xcd = attn_xcd_flag(B, H) | ((flags & DEVIAS_ATTN_Q_PRESCALED) ? ATTN_QPRE : 0);
        devias_count(DEVIAS_CNT_MHSA_FWD_BF16);
        if (xcd & ATTN_QPRE) devias_count(DEVIAS_CNT_MHSA_QPRE);
#define FWD_GRID(QB) (xcd & 1) ? dim3(cdiv(N, QB) * H * B) : dim3(cdiv(N, QB), H, B)
        if (drop) hipLaunchKernelGGL((mhsa_fwd32_bf16_kernel<4, true>), FWD_GRID(128), dim3(256), 0, st, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, xcd, dp);
        else if (cfg == 6) hipLaunchKernelGGL((mhsa_fwd_bf16_kernel<2, 4, true>), FWD_GRID(128), dim3(256), 0, st, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, xcd);
        else if (cfg == 7) hipLaunchKernelGGL((mhsa_fwd32_bf16_kernel<2>), FWD_GRID(64), dim3(128), 0, st, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, xcd, dp);
        // default (measured at B = 32, H = 12, N = 1568, same box): 32x32x16 kernel 302 us against 334-345 us for the 16x16x32 kernel (cfg 6)
        else hipLaunchKernelGGL((mhsa_fwd32_bf16_kernel<4>), FWD_GRID(128), dim3(256), 0, st, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, xcd, dp);
#undef FWD_GRID
    }
    else if (dtype == DEVIAS_F32) {
        devias_count(DEVIAS_CNT_MHSA_FWD_F32);
        if (drop) hipLaunchKernelGGL(mhsa_fwd_f32_kernel<true>, dim3(cdiv(N, 128), H, B), dim3(128), 0, st, (const float*)qkv, (float*)o, lse, N, H, scale, dp);
        else hipLaunchKernelGGL(mhsa_fwd_f32_kernel<false>, dim3(cdiv(N, 128), H, B), dim3(128), 0, st, (const float*)qkv, (float*)o, lse, N, H, scale, dp);
    } else return devias_set_error(DEVIAS_EINVAL, "devias_mhsa_fwd: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_mhsa_fwd");
    return DEVIAS_OK;
}
extern "C" int devias_mhsa_fwd(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                               int32_t dtype, void* stream) {
    return mhsa_fwd_impl(qkv, o, lse, B, N, H, scale, dtype, 1.0f, 0, stream);
}
extern "C" int devias_mhsa_fwd_flags(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                                     int32_t dtype, int32_t flags, void* stream) {
    return mhsa_fwd_impl(qkv, o, lse, B, N, H, scale, dtype, 1.0f, 0, stream, flags);
}
extern "C" int devias_mhsa_fwd_dropout(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                                       int32_t dtype, float keep, uint64_t seed, void* stream) {
    DEVIAS_REQUIRE(keep > 0.f && keep <= 1.f, "devias_mhsa_fwd_dropout: keep must be in (0, 1]");
    return mhsa_fwd_impl(qkv, o, lse, B, N, H, scale, dtype, keep, seed, stream);
}

// (ABI 140 had a single-pass backward that needed a workspace; it was removed in ABI 150 -- see DESIGN.md -- and the query stays for hosts
// written against the older header: nothing is needed any more)
// Row statistics for the one-wave-per-SIMD dK / dV kernel (attn_bwd1w.hip): [B, H, Npad / 32, 2, 32] fp32 (per 32-query slice -lse * log2 e | -delta), Npad = N
// rounded up to a slice; written by the dQ kernel, streamed by LDS-DMA by the dK / dV kernel.
static inline int attn_npad(int N) { return (N + 31) & ~31; }
static inline int64_t attn_stat_bytes(int B, int N, int H) { return (((int64_t)B * H * 2 * attn_npad(N) * 4) + 255) & ~(int64_t)255; }
extern "C" int64_t devias_mhsa_bwd_workspace_bytes(int32_t B, int32_t N, int32_t H) { return attn_stat_bytes(B, N, H); }
int devias_attn_dkdv1w_launch(const void* qkv, const void* d_o, const float* stat, void* dqkv, int B, int N, int Npad, int H, float scale, int xcd_flag,
                              int persistent, hipStream_t st);      // attn_bwd1w.hip
// (A one-wave-per-SIMD dQ kernel was built the same way and is SLOWER than the three-waves-per-SIMD kernel below after its vector-instruction diet -- 445 us
// against 337 per layer: tools/exp/attn_bwd1w_dq.hip.txt, profiles/r5_dkdv1w_development.txt.  One wave overlaps its own MFMAs and vector instructions only inside
// the MFMA's shadow; three waves overlap each other's.  The dK / dV kernel wins as one wave because its 128 accumulator registers leave no room for a second.)
// true = devias_mhsa_bwd* runs the one-wave-per-SIMD dK / dV kernel for this call (bf16, no attention dropout, option attn_dkdv != 0, room for the statistics)
static inline bool attn_use_dkdv1w(int dtype, float keep) { return dtype == DEVIAS_BF16 && !(keep < 1.0f) && attn_knobs().dkdv != 0; }


static int mhsa_bwd_impl(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                         int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, float keep, uint64_t seed, void* stream,
                         float* part_q = nullptr, float* part_v = nullptr, float* stat = nullptr, int32_t flags = 0) {
    hipStream_t st = (hipStream_t)stream;
    DropP dp{};
    const bool drop = drop_params(keep, seed, dp);
    DEVIAS_REQUIRE(qkv && o && d_o && lse && delta && dqkv && B > 0 && N > 0 && H > 0, "devias_mhsa_bwd: bad args");
    DEVIAS_REQUIRE(aligned16(qkv) && aligned16(o) && aligned16(d_o) && aligned16(dqkv), "devias_mhsa_bwd: unaligned pointer");
    DEVIAS_REQUIRE(H <= 65535 && B <= 65535, "devias_mhsa_bwd: H and B must be <= 65535");
    DEVIAS_REQUIRE((flags & ~DEVIAS_ATTN_Q_PRESCALED) == 0 && (flags == 0 || dtype == DEVIAS_BF16), "devias_mhsa_bwd: bad flags %d (DEVIAS_ATTN_Q_PRESCALED: bf16 only)", flags);
    if (dtype == DEVIAS_BF16) {
        const int cfg = attn_knobs().cfg;
        const int xcd = attn_xcd_flag(B, H) | ((flags & DEVIAS_ATTN_Q_PRESCALED) ? ATTN_QPRE : 0);
        devias_count(DEVIAS_CNT_MHSA_BWD_BF16);
        if (xcd & ATTN_QPRE) devias_count(DEVIAS_CNT_MHSA_QPRE);
        // dK / dV by the one-wave-per-SIMD kernel when the caller gave room for the row statistics it streams (option attn_dkdv = 0: the two-waves-per-SIMD
        // kernel, which also serves attention dropout)
        const bool w1 = stat && attn_use_dkdv1w(dtype, keep);
        DEVIAS_REQUIRE(!(w1 && part_v), "devias_mhsa_bwd: the one-wave-per-SIMD dK / dV kernel emits no v_bias partials (internal)");
        float* const stat_w = w1 ? stat : nullptr;
        const int npad = attn_npad(N);
#define DQ_ARGS (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, xcd, dp, part_q, stat_w, npad
#define BWD_GRID(QB) (xcd & 1) ? dim3(cdiv(N, QB) * H * B) : dim3(cdiv(N, QB), H, B)
        if (drop) hipLaunchKernelGGL((mhsa_bwd_dq_bf16_kernel<2, 4, true>), BWD_GRID(128), dim3(256), 0, st, DQ_ARGS);
        else if (cfg == 1 && !part_q) hipLaunchKernelGGL((mhsa_bwd_dq_bf16_kernel<4, 2>), BWD_GRID(128), dim3(128), 0, st, DQ_ARGS);
        else if (cfg == 2 && !part_q) hipLaunchKernelGGL((mhsa_bwd_dq_bf16_kernel<4, 4>), BWD_GRID(256), dim3(256), 0, st, DQ_ARGS);
        else if (cfg == 3 && !part_q) hipLaunchKernelGGL((mhsa_bwd_dq_bf16_kernel<2, 2>), BWD_GRID(64), dim3(128), 0, st, DQ_ARGS);
        else hipLaunchKernelGGL((mhsa_bwd_dq_bf16_kernel<2, 4>), BWD_GRID(128), dim3(256), 0, st, DQ_ARGS);
#undef DQ_ARGS
        DEVIAS_CHECK_LAUNCH("devias_mhsa_bwd(dq)");
        if (w1) {
            const int rc = devias_attn_dkdv1w_launch(qkv, d_o, stat, dqkv, B, N, npad, H, scale, xcd, attn_knobs().dkdv == 2, st);
            if (rc != DEVIAS_OK) return rc;
        } else {
            devias_count(DEVIAS_CNT_DKDV2W);
            if (drop) hipLaunchKernelGGL(mhsa_bwd_dkdv_bf16_kernel<true>, BWD_GRID(128), dim3(256), 0, st, (const bf16*)qkv, (const bf16*)d_o,
                                         lse, delta, (bf16*)dqkv, N, H, scale, xcd, dp, part_v);
            else hipLaunchKernelGGL(mhsa_bwd_dkdv_bf16_kernel<false>, BWD_GRID(128), dim3(256), 0, st, (const bf16*)qkv, (const bf16*)d_o,
                                    lse, delta, (bf16*)dqkv, N, H, scale, xcd, dp, part_v);
        }
#undef BWD_GRID
        DEVIAS_CHECK_LAUNCH("devias_mhsa_bwd(dkdv)");
    } else if (dtype == DEVIAS_F32) {
        devias_count(DEVIAS_CNT_MHSA_BWD_F32);
        if (drop) hipLaunchKernelGGL(mhsa_bwd_dq_f32_kernel<true>, dim3(cdiv(N, 128), H, B), dim3(128), 0, st, (const float*)qkv, (const float*)o,
                                     (const float*)d_o, lse, delta, (float*)dqkv, N, H, scale, dp);
        else hipLaunchKernelGGL(mhsa_bwd_dq_f32_kernel<false>, dim3(cdiv(N, 128), H, B), dim3(128), 0, st, (const float*)qkv, (const float*)o,
                                (const float*)d_o, lse, delta, (float*)dqkv, N, H, scale, dp);
        DEVIAS_CHECK_LAUNCH("devias_mhsa_bwd(dq)");
        if (drop) hipLaunchKernelGGL(mhsa_bwd_dkdv_f32_kernel<true>, dim3(cdiv(N, 64), H, B), dim3(128), 0, st, (const float*)qkv, (const float*)d_o,
                                     lse, delta, (float*)dqkv, N, H, scale, dp);
        else hipLaunchKernelGGL(mhsa_bwd_dkdv_f32_kernel<false>, dim3(cdiv(N, 64), H, B), dim3(128), 0, st, (const float*)qkv, (const float*)d_o,
                                lse, delta, (float*)dqkv, N, H, scale, dp);
        DEVIAS_CHECK_LAUNCH("devias_mhsa_bwd(dkdv)");
    } else return devias_set_error(DEVIAS_EINVAL, "devias_mhsa_bwd: bad dtype %d", dtype);
    return DEVIAS_OK;
}
extern "C" int devias_mhsa_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                               int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, void* ws, void* stream) {
    DEVIAS_REQUIRE(!ws || aligned16(ws), "devias_mhsa_bwd: unaligned workspace");
    return mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, 1.0f, 0, stream, nullptr, nullptr, (float*)ws);
}
extern "C" int devias_mhsa_bwd_flags(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                                     int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, void* ws, int32_t flags, void* stream) {
    DEVIAS_REQUIRE(!ws || aligned16(ws), "devias_mhsa_bwd: unaligned workspace");
    return mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, 1.0f, 0, stream, nullptr, nullptr, (float*)ws, flags);
}
extern "C" int devias_mhsa_bwd_dropout(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                                       int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, float keep, uint64_t seed, void* stream) {
    DEVIAS_REQUIRE(keep > 0.f && keep <= 1.f, "devias_mhsa_bwd_dropout: keep must be in (0, 1]");
    return mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, stream);
}

// Backward + the q_bias / v_bias gradients (the column sums of the dQ and dV thirds of dqkv over all B * N rows; modeling_slot.py:97-99).  bf16: the two kernels
// emit one [H * 64] partial per (batch entry, 128-row block) from their fp32 accumulators (bias_partials above) and the fixed-order second stage sums the
// B * ceil(N / 128) partials -- no pass over the stored tensor.  fp32 (parity mode): the plain backward followed by two column-sum passes.  ws_q / ws_v: scratch of
// devias_mhsa_bwd_bias_workspace_bytes() bytes each.  keep < 1: with the attention dropout of devias_mhsa_bwd_dropout.  (ABI 162)
static inline int64_t attn_bias_part_bytes(int B, int N, int H) { return (((int64_t)B * cdiv(N, 128) * H * 64 * 4) + 255) & ~(int64_t)255; }
extern "C" int64_t devias_mhsa_bwd_bias_workspace_bytes(int32_t B, int32_t N, int32_t H) {
    const int64_t a = attn_bias_part_bytes(B, N, H) + attn_stat_bytes(B, N, H), c = devias_colsum_workspace_bytes(B * N, H * 64);
    return a > c ? a : c;
}
extern "C" int32_t devias_mhsa_bwd_bias_dv_from_do(int32_t dtype, float keep) { return attn_use_dkdv1w(dtype, keep) ? 1 : 0; }
static int mhsa_bwd_bias_impl(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                              float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, void* stream, int32_t flags);
extern "C" int devias_mhsa_bwd_bias(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                                    float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, void* stream) {
    return mhsa_bwd_bias_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, dbq, dbv, ws_q, ws_v, stream, 0);
}
extern "C" int devias_mhsa_bwd_bias_flags(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                                          float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, int32_t flags, void* stream) {
    return mhsa_bwd_bias_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, dbq, dbv, ws_q, ws_v, stream, flags);
}
static int mhsa_bwd_bias_impl(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                              float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, void* stream, int32_t flags) {
    DEVIAS_REQUIRE(dbq && ws_q && ws_v, "devias_mhsa_bwd_bias: null bias-gradient / workspace pointer");
    DEVIAS_REQUIRE(keep > 0.f && keep <= 1.f, "devias_mhsa_bwd_bias: keep must be in (0, 1]");
    DEVIAS_REQUIRE(!(flags != 0 && keep < 1.f), "devias_mhsa_bwd_bias_flags: DEVIAS_ATTN_Q_PRESCALED is not offered with attention dropout (no forward entry point takes both)");
    const int D = H * 64;
    const int64_t es = dtype == DEVIAS_BF16 ? 2 : 4;
    if (attn_use_dkdv1w(dtype, keep)) {
        // One-wave-per-SIMD dK / dV kernel.  Its row statistics live behind the q partials in ws_q.  It emits no v partials: without dropout every softmax row sums
        // to one, so sum_keys dV = sum_keys sum_queries P dO = sum_queries dO -- the v_bias gradient IS the column sum of d_o.  A caller that has those column sums
        // from the producer of d_o (the projection's dgrad GEMM: devias_gemm's colsum epilogue, csrc/regions.hip) passes dbv = NULL; otherwise one pass over d_o here.
        float* stat = reinterpret_cast<float*>(reinterpret_cast<char*>(ws_q) + attn_bias_part_bytes(B, N, H));
        const int rc = mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, stream, ws_q, nullptr, stat, flags);
        if (rc != DEVIAS_OK) return rc;
        const int r1 = devias_colsum_finish(ws_q, B * cdiv(N, 128), D, dbq, 0.f, (hipStream_t)stream);
        if (r1 != DEVIAS_OK || !dbv) return r1;
        return devias_colsum(d_o, dtype, B * N, D, D, dbv, 0.f, ws_v, stream);
    }
    DEVIAS_REQUIRE(dbv, "devias_mhsa_bwd_bias: dbv may be NULL only where devias_mhsa_bwd_bias_dv_from_do() says so");
    if (dtype == DEVIAS_BF16 && attn_knobs().bias_fused) {
        const int rc = mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, stream, ws_q, ws_v, nullptr, flags);
        if (rc != DEVIAS_OK) return rc;
        const int rows = B * cdiv(N, 128);
        const int r1 = devias_colsum_finish(ws_q, rows, D, dbq, 0.f, (hipStream_t)stream);
        return r1 != DEVIAS_OK ? r1 : devias_colsum_finish(ws_v, rows, D, dbv, 0.f, (hipStream_t)stream);
    }
    const int rc = mhsa_bwd_impl(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, keep, seed, stream, nullptr, nullptr, nullptr, flags);
    if (rc != DEVIAS_OK) return rc;
    const int r1 = devias_colsum(dqkv, dtype, B * N, D, 3 * D, dbq, 0.f, ws_q, stream);
    return r1 != DEVIAS_OK ? r1 : devias_colsum(static_cast<const char*>(dqkv) + (int64_t)2 * D * es, dtype, B * N, D, 3 * D, dbv, 0.f, ws_v, stream);
}
