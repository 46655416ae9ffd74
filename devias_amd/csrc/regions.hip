// Fused regions: ONE C-ABI call enqueues the whole kernel sequence of a block of the step (host-side sequencing only; every kernel is
// the same launch the per-kernel entry points make, in the same order, with the same split / workspace policy -- results are bitwise
// those of the per-kernel path).  Why: the step is ~750 launches; issued one ctypes hop at a time the Python host needs 30-43 ms per
// 55 ms device step, and eight such processes share one host on an 8-GPU node.  With these entry points a step is ~40 host -> library calls.
//
//   devias_encoder_block_fwd/bwd   Block.forward of model/modeling_slot.py:142-152 (LN -> QKV -> MHSA -> proj+res -> LN -> fc1+GELU -> fc2+res)
//   devias_agg_block_fwd/bwd       final LayerNorm + AggregationBlock, folded slot attention (modeling_slot.py:373,381; agg_block/agg_block.py:120-139;
//                                  agg_block/attention.py:32-40,120-141)
//   devias_head_fwd/bwd            shared head + MaskPredictor (modeling_slot.py:392-393, 194-216)
// Memory: everything is caller-owned.  `save` arenas hold what backward needs (opaque layout, sized by the *_save_bytes functions),
// `scratch` holds backward temporaries, `ws` is the small fp32 workspace the individual kernels use (split-K slabs, partial sums).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include "roctx_shim.h"

namespace {

inline int64_t al256(int64_t n) { return (n + 255) & ~(int64_t)255; }
inline int esize(int dtype) { return dtype == DEVIAS_BF16 ? 2 : 4; }

// ---- the split policies of the Python host (devias_amd/ops.py: gemm / wgrad / auto_split_k), restated; tests/test_regions_cpu.py keeps them equal ----
int small_m_split(int M, int N, int K, int trans_a) {
    // (measured in the step, bench.py interleaved: never splitting these costs +2.2 ms, splitting only from K = 1024 up +0.7 ms)
    if (!(M <= 256 && K >= 512 && !trans_a)) return 1;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    int s = K / 128;
    const int t = (256 + tiles - 1) / tiles;
    if (t < s) s = t;
    return s > 1 ? s : 1;
}
int wgrad_split(int Nout, int Kin, int Mrows, int dtype) {      // C is [Nout, Kin], the reduction runs over Mrows
    const int bk = dtype == DEVIAS_BF16 ? 64 : 16;
    const int cus = devias_policy_gemm_cus();                   // 256 on MI355X; fewer when CUs are reserved for a concurrent kernel (gemm_reserve_cus)
    int tiles, slots;
    if (bk == 64 && Nout % 256 == 0 && Kin % 128 == 0) { tiles = (Nout / 256) * (Kin / 128); slots = 2 * cus; }     // (tiles x splits) fills ONE round of the 256^2 kernel
    else { tiles = cdiv(Nout, 128) * cdiv(Kin, 128); slots = 3 * cus; }
    if (tiles >= slots || Mrows < 8 * bk) return 1;
    int s = slots / tiles;
    if (Mrows / (4 * bk) < s) s = Mrows / (4 * bk);
    if (s > 64) s = 64;
    return s > 1 ? s : 1;
}

struct Ctx {
    int dtype;
    float* ws; int64_t ws_bytes;
    void* sk_ws; int64_t sk_ws_bytes;
    void* st;
};

// C[M,N] = epi(op(A) op(B)) with the Python host's automatic choices (small-M split-K)
struct Epi {
    const float* bias = nullptr; int act = DEVIAS_ACT_NONE;
    const void* aux_in = nullptr; void* aux_out = nullptr;
    const void* res = nullptr; int res_mod = 0;
    float* colsum = nullptr; float colsum_beta = 0.f;
    const float* row_scale = nullptr; int rows_per_scale = 0;
};
int gemm(const Ctx& c, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b, const Epi& e,
         int c_f32 = 0, float beta = 0.f, int split_k = 1) {
    devias_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = N;
    a.trans_a = trans_a; a.trans_b = trans_b; a.dtype = c.dtype; a.c_f32 = c_f32;
    a.bias = e.bias; a.act = e.act; a.aux_in = e.aux_in; a.aux_out = e.aux_out; a.ld_aux = N;
    a.res = e.res; a.ldr = N; a.res_mod = e.res_mod; a.beta = beta;
    a.row_scale = e.row_scale; a.rows_per_scale = e.rows_per_scale;
    if (split_k == 1) split_k = small_m_split(M, N, K, trans_a);
    if (e.colsum) { split_k = 1; a.colsum = e.colsum; a.colsum_beta = e.colsum_beta; a.ws = c.ws; }
    a.split_k = split_k;
    if (split_k > 1) {
        if (devias_gemm_workspace_bytes(M, N, split_k) > c.ws_bytes) return devias_set_error(DEVIAS_EINVAL, "fused region: workspace too small for a split-K GEMM (%d x %d x %d)", M, N, split_k);
        a.ws = c.ws;
    }
    return devias_gemm(&a, c.st);
}
// dW[Nout,Kin] (fp32) = beta * dW + dY[M,Nout]^T X[M,Kin]
int wgrad(const Ctx& c, const void* dY, const void* X, float* dW, int M, int Nout, int Kin, float beta = 0.f) {
    Epi e;
    return gemm(c, dY, X, dW, Nout, Kin, M, Nout, Kin, 1, 1, e, 1, beta, wgrad_split(Nout, Kin, M, c.dtype));
}
int gemm_batched(const Ctx& c, const void* A, const void* B, void* C, int c_f32, int M, int N, int K, int lda, int ldb, int ldc, int64_t sa, int64_t sb, int64_t sc,
                 int batch, int trans_a, int trans_b) {
    devias_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.trans_a = trans_a; a.trans_b = trans_b; a.dtype = c.dtype; a.c_f32 = c_f32; a.split_k = 1;
    a.batch = batch; a.stride_a = sa; a.stride_b = sb; a.stride_c = sc;
    return devias_gemm(&a, c.st);
}
int ln_fwd(const Ctx& c, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int M, int D, float eps) {
    return devias_layernorm_fwd(x, g, b, y, mean, rstd, M, D, eps, c.dtype, c.st);
}
int ln_bwd(const Ctx& c, const void* dy, const void* x, const float* g, const float* mean, const float* rstd, const void* dres, void* dx, float* dg, float* db,
           float beta_acc, float* dx_colsum, int M, int D) {
    return devias_layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, beta_acc, dx_colsum, M, D, c.dtype, c.ws, c.st);
}
int colsum(const Ctx& c, const void* x, int M, int N, float* out, float beta = 0.f) { return devias_colsum(x, c.dtype, M, N, N, out, beta, c.ws, c.st); }

#define RUN(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
int64_t gemm_ws(int M, int N, int K, int trans_a) { return devias_gemm_workspace_bytes(M, N, small_m_split(M, N, K, trans_a)); }
int64_t wgrad_ws(int Nout, int Kin, int M, int dtype) { return devias_gemm_workspace_bytes(Nout, Kin, wgrad_split(Nout, Kin, M, dtype)); }
int64_t colsum_gemm_ws(int M, int N) { return max64((int64_t)cdiv(M, 128) * N * 4, devias_colsum_workspace_bytes(M, N)); }

// ================================================ encoder block ==================================================================
struct BlockSave {
    char *u, *qkv, *o, *x1, *u2, *hpre, *hact;
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
    int64_t bytes;
    BlockSave(void* base, int B, int N, int D, int H, int hid, int dtype) {
        const int64_t M = (int64_t)B * N, es = esize(dtype);
        int64_t off = 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(base);          // (integer arithmetic: the size queries lay the arena out at address 0)
        auto take = [&](int64_t n) { char* r = reinterpret_cast<char*>(p + (uintptr_t)off); off += al256(n); return r; };
        u = take(M * D * es); qkv = take(M * 3 * D * es); o = take(M * D * es); x1 = take(M * D * es); u2 = take(M * D * es);
        hpre = take(M * hid * es); hact = take(M * hid * es);
        mean1 = (float*)take(M * 4); rstd1 = (float*)take(M * 4); mean2 = (float*)take(M * 4); rstd2 = (float*)take(M * 4);
        lse = (float*)take((int64_t)B * H * N * 4);
        bytes = off;
    }
};
struct BlockScratch {
    char *big, *small, *dx1, *g;      // big: dhpre [M,hid] then dqkv [M,3D]; small: du2, d_o, du in turn [M,D]; g: the row-scaled copy of a branch gradient (stochastic depth)
    float* delta;
    // private partial areas of the kernels whose second stage the region runs as ONE launch at its end (common.h: deferred final reductions):
    // the two LayerNorm backwards, the dfc2 GEMM's column-sum epilogue (db1), and the column sums for db2 / dbp (stochastic depth) / dq_bias / dv_bias
    float *p_ln2, *p_ln1, *p_db1, *p_b2, *p_bp, *p_q, *p_v;
    int64_t bytes;
    BlockScratch(void* base, int B, int N, int D, int H, int hid, int dtype) {
        const int64_t M = (int64_t)B * N, es = esize(dtype);
        int64_t off = 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(base);          // (integer arithmetic: the size queries lay the arena out at address 0)
        auto take = [&](int64_t n) { char* r = reinterpret_cast<char*>(p + (uintptr_t)off); off += al256(n); return r; };
        big = take(M * (hid > 3 * D ? hid : 3 * D) * es); small = take(M * D * es); dx1 = take(M * D * es); g = take(M * D * es);
        delta = (float*)take((int64_t)B * H * N * 4);
        const int64_t lnw = devias_layernorm_bwd_workspace_bytes((int)M, D), csw = devias_colsum_workspace_bytes((int)M, 3 * D);
        p_ln2 = (float*)take(lnw); p_ln1 = (float*)take(lnw);
        p_db1 = (float*)take(max64((int64_t)cdiv(M, 128) * hid * 4, devias_colsum_workspace_bytes((int)M, hid)));
        p_b2 = (float*)take(csw); p_bp = (float*)take(csw);
        const int64_t abw = max64(csw, devias_mhsa_bwd_bias_workspace_bytes(B, N, H));
        p_q = (float*)take(abw); p_v = (float*)take(abw);
        bytes = off;
    }
};
// forward ran the qkv GEMM on the caller's q-scaled copies (ABI 167): bf16, both pointers given
inline bool block_qpre(const devias_block_args* a) { return a->dtype == DEVIAS_BF16 && a->WqkvS && a->qkv_biasS; }
int block_check(const devias_block_args* a, const char* who) {
    DEVIAS_REQUIRE(a, "%s: null args", who);
    DEVIAS_REQUIRE(a->B > 0 && a->N > 0 && a->D > 0 && a->H > 0 && a->hidden > 0 && a->D == a->H * 64, "%s: bad dims (B=%d N=%d D=%d H=%d hidden=%d; head dim must be 64)", who,
                   a->B, a->N, a->D, a->H, a->hidden);
    DEVIAS_REQUIRE(a->dtype == DEVIAS_BF16 || a->dtype == DEVIAS_F32, "%s: bad dtype %d", who, a->dtype);
    DEVIAS_REQUIRE(a->n1w && a->n1b && a->n2w && a->n2b && a->Wqkv && a->Wp && a->W1 && a->W2 && a->qkv_bias && a->pb && a->b1 && a->b2, "%s: null parameter", who);
    DEVIAS_REQUIRE((a->WqkvS != nullptr) == (a->qkv_biasS != nullptr) && (!a->WqkvS || a->dtype == DEVIAS_BF16), "%s: WqkvS and qkv_biasS go together (bf16 only)", who);
    DEVIAS_REQUIRE(a->save && a->ws && aligned16(a->save) && aligned16(a->ws), "%s: save / ws must be 16-byte aligned, non-null", who);
    DEVIAS_REQUIRE(a->ws_bytes >= devias_encoder_block_workspace_bytes(a->B, a->N, a->D, a->H, a->hidden, a->dtype), "%s: workspace too small", who);
    return DEVIAS_OK;
}

}  // namespace

extern "C" int64_t devias_encoder_block_save_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype) {
    return BlockSave(nullptr, B, N, D, H, hidden, dtype).bytes;
}
extern "C" int64_t devias_encoder_block_scratch_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype) {
    return BlockScratch(nullptr, B, N, D, H, hidden, dtype).bytes;
}
extern "C" int64_t devias_encoder_block_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype) {
    const int M = B * N;
    int64_t w = devias_layernorm_bwd_workspace_bytes(M, D);
    w = max64(w, colsum_gemm_ws(M, hidden));
    w = max64(w, devias_colsum_workspace_bytes(M, 3 * D));
    w = max64(w, max64(gemm_ws(M, 3 * D, D, 0), max64(gemm_ws(M, hidden, D, 0), max64(gemm_ws(M, D, hidden, 0), gemm_ws(M, D, 3 * D, 0)))));
    w = max64(w, max64(wgrad_ws(D, hidden, M, dtype), max64(wgrad_ws(hidden, D, M, dtype), max64(wgrad_ws(D, D, M, dtype), wgrad_ws(3 * D, D, M, dtype)))));
    return al256(w) + 256;
}

extern "C" int devias_encoder_block_fwd(const devias_block_args* a, const void* x, void* x2, void* stream) {
    RUN(block_check(a, "devias_encoder_block_fwd"));
    DEVIAS_REQUIRE(x && x2, "devias_encoder_block_fwd: null activation");
    const int B = a->B, N = a->N, D = a->D, H = a->H, hid = a->hidden, M = B * N;
    const Ctx c{a->dtype, a->ws, a->ws_bytes, a->sk_ws, a->sk_ws_bytes, stream};
    const BlockSave s(a->save, B, N, D, H, hid, a->dtype);
    devias_range r("encoder_block_fwd");
    RUN(ln_fwd(c, x, a->n1w, a->n1b, s.u, s.mean1, s.rstd1, M, D, a->eps));
    // q pre-multiplied by scale * log2 e where the caller keeps such a copy of Wqkv / qkv_bias (ABI 167): the arena's q third is then q', rounded once, and the attention
    // kernels of both directions multiply the same bf16 operands (devias_mhsa_fwd_flags)
    const bool qpre = block_qpre(a);
    { Epi e; e.bias = qpre ? a->qkv_biasS : a->qkv_bias; RUN(gemm(c, s.u, qpre ? a->WqkvS : a->Wqkv, s.qkv, M, 3 * D, D, D, D, 0, 0, e)); }            // [M, 3D] == [B,N,3,H,64]
    RUN(devias_mhsa_fwd_flags(s.qkv, s.o, s.lse, B, N, H, 0.125f, a->dtype, qpre ? DEVIAS_ATTN_Q_PRESCALED : 0, stream));
    { Epi e; e.bias = a->pb; e.res = x; e.row_scale = a->ds1; e.rows_per_scale = N; RUN(gemm(c, s.o, a->Wp, s.x1, M, D, D, D, D, 0, 0, e)); }      // x + drop_path(proj(.))
    RUN(ln_fwd(c, s.x1, a->n2w, a->n2b, s.u2, s.mean2, s.rstd2, M, D, a->eps));
    { Epi e; e.bias = a->b1; e.act = DEVIAS_ACT_GELU; e.aux_out = s.hpre; RUN(gemm(c, s.u2, a->W1, s.hact, M, hid, D, D, D, 0, 0, e)); }
    { Epi e; e.bias = a->b2; e.res = s.x1; e.row_scale = a->ds2; e.rows_per_scale = N; RUN(gemm(c, s.hact, a->W2, x2, M, D, hid, hid, hid, 0, 0, e)); }  // x1 + drop_path(mlp(.))
    return DEVIAS_OK;
}

extern "C" int devias_encoder_block_bwd(const devias_block_args* a, const void* x, const void* dx2, void* dx, const devias_block_grads* g,
                                        void* scratch, int64_t scratch_bytes, void* stream) {
    RUN(block_check(a, "devias_encoder_block_bwd"));
    DEVIAS_REQUIRE(x && dx2 && dx && g && scratch && aligned16(scratch), "devias_encoder_block_bwd: null / unaligned argument");
    DEVIAS_REQUIRE(g->dn1w && g->dn1b && g->dWqkv && (g->dbqkv || (g->dbq && g->dbv)) && g->dWp && g->dbp && g->dn2w && g->dn2b && g->dW1 && g->db1 && g->dW2 && g->dx_colsum &&
                   (g->db2 || g->db2_done), "devias_encoder_block_bwd: null gradient destination");
    const int B = a->B, N = a->N, D = a->D, H = a->H, hid = a->hidden, M = B * N;
    DEVIAS_REQUIRE(scratch_bytes >= devias_encoder_block_scratch_bytes(B, N, D, H, hid, a->dtype), "devias_encoder_block_bwd: scratch too small");
    const Ctx c{a->dtype, a->ws, a->ws_bytes, a->sk_ws, a->sk_ws_bytes, stream};
    const BlockSave s(a->save, B, N, D, H, hid, a->dtype);
    const BlockScratch t(scratch, B, N, D, H, hid, a->dtype);
    devias_range r("encoder_block_bwd");
    // transposed weight copies for the dgrad GEMMs: bf16 only (the fp32 kernels stage through registers: no gain), option gemm_wt
    const bool wt = a->dtype == DEVIAS_BF16 && devias_policy_gemm_wt() != 0;
    const void *WqkvT = wt ? a->WqkvT : nullptr, *WpT = wt ? a->WpT : nullptr, *W1T = wt ? a->W1T : nullptr, *W2T = wt ? a->W2T : nullptr;
    // The second stages of this region's partial reductions (two LayerNorm parameter reduces, the column-sum finals of db1 / db2 / dbp / dq_bias / dv_bias:
    // 5-7 launches of 6-12 us) run as ONE launch at the end: each producer gets a private partial area in the scratch arena (Ctx with that `ws`).
    DeviasDeferList dl; dl.n = 0;
    struct Collect { Collect(DeviasDeferList* l) { devias_defer_slot() = l; } ~Collect() { devias_defer_slot() = nullptr; } } collecting(&dl);
    auto with_ws = [&](float* area) { Ctx k = c; k.ws = area; return k; };
    // ---- MLP branch (g2 = gradient of the branch output: dx2 scaled by the per-sample stochastic-depth factor, if any)
    const void* g2 = dx2;
    if (a->ds2) {
        RUN(devias_row_scale_colsum(dx2, a->ds2, N, t.g, a->dtype, M, D, g->db2, 0.f, t.p_b2, (hipStream_t)stream));      // the rescaled copy and its column sums in one pass
        g2 = t.g;
    } else if (!g->db2_done) {
        RUN(colsum(with_ws(t.p_b2), dx2, M, D, g->db2));                       // fc2 bias gradient (the caller had no ready-made column sums of dx2)
    }
    RUN(wgrad(c, g2, s.hact, g->dW2, M, D, hid));
    // (the four dgrad GEMMs: on the caller's transposed weight copies where it keeps them -- both operands k-contiguous -- else the weight read k-strided; same bits)
    { Epi e; e.act = DEVIAS_ACT_DGELU; e.aux_in = s.hpre; e.colsum = g->db1;                                                                                   // (g2 W2) * gelu'(pre); db1 = colsum
      if (W2T) RUN(gemm(with_ws(t.p_db1), g2, W2T, t.big, M, hid, D, D, D, 0, 0, e)); else RUN(gemm(with_ws(t.p_db1), g2, a->W2, t.big, M, hid, D, D, hid, 0, 1, e)); }
    RUN(wgrad(c, t.big, s.u2, g->dW1, M, hid, D));
    { Epi e; if (W1T) RUN(gemm(c, t.big, W1T, t.small, M, D, hid, hid, hid, 0, 0, e)); else RUN(gemm(c, t.big, a->W1, t.small, M, D, hid, hid, D, 0, 1, e)); }   // du2
    // + residual gradient; dbp = colsum(dx1) -- unless stochastic depth rescales dx1 first: then the column sum of the rescaled copy below is dbp (two deferred
    // jobs must not share a destination: they run side by side)
    RUN(ln_bwd(with_ws(t.p_ln2), t.small, s.x1, a->n2w, s.mean2, s.rstd2, dx2, t.dx1, g->dn2w, g->dn2b, 0.f, a->ds1 ? nullptr : g->dbp, M, D));
    // ---- attention branch
    const void* g1 = t.dx1;
    if (a->ds1) {
        RUN(devias_row_scale_colsum(t.dx1, a->ds1, N, t.g, a->dtype, M, D, g->dbp, 0.f, t.p_bp, (hipStream_t)stream));
        g1 = t.g;
    }
    RUN(wgrad(c, g1, s.o, g->dWp, M, D, D));
    // v_bias gradient: the column sum of d_o where the attention backward says so (softmax rows sum to one: sum_keys dV = sum_queries dO), from this GEMM's epilogue
    const bool dv_from_do = g->dbq && g->dbv && devias_mhsa_bwd_bias_dv_from_do(a->dtype, 1.0f);
    const int aflags = block_qpre(a) ? DEVIAS_ATTN_Q_PRESCALED : 0;      // (the arena's q third is q' = q * scale * log2 e: what forward decided from the same struct)
    { Epi e; if (dv_from_do) e.colsum = g->dbv;                                                                                               // d_o (p_v: the attention backward does not use it then)
      const Ctx k = dv_from_do ? with_ws(t.p_v) : c;
      if (WpT) RUN(gemm(k, g1, WpT, t.small, M, D, D, D, D, 0, 0, e)); else RUN(gemm(k, g1, a->Wp, t.small, M, D, D, D, D, 0, 1, e)); }
    if (g->dbq && g->dbv)                                                   // dqkv + the q_bias / v_bias gradients (each to its own destination) from the same two kernels
        RUN(devias_mhsa_bwd_bias_flags(s.qkv, s.o, t.small, s.lse, t.delta, t.big, B, N, H, 0.125f, a->dtype, 1.0f, 0, g->dbq, dv_from_do ? nullptr : g->dbv, t.p_q, t.p_v, aflags, stream));
    else
        RUN(devias_mhsa_bwd_flags(s.qkv, s.o, t.small, s.lse, t.delta, t.big, B, N, H, 0.125f, a->dtype, t.p_q, aflags, stream));                          // dqkv
    RUN(wgrad(c, t.big, s.u, g->dWqkv, M, 3 * D, D));
    if (g->dbq && g->dbv) {
    } else {
        RUN(colsum(with_ws(t.p_q), t.big, M, 3 * D, g->dbqkv));
    }
    { Epi e; if (WqkvT) RUN(gemm(c, t.big, WqkvT, t.small, M, D, 3 * D, 3 * D, 3 * D, 0, 0, e)); else RUN(gemm(c, t.big, a->Wqkv, t.small, M, D, 3 * D, 3 * D, D, 0, 1, e)); }   // du
    RUN(ln_bwd(with_ws(t.p_ln1), t.small, x, a->n1w, s.mean1, s.rstd1, t.dx1, dx, g->dn1w, g->dn1b, 0.f, g->dx_colsum, M, D));
    RUN(devias_flush_deferred(&dl, (hipStream_t)stream));
    return DEVIAS_OK;
}

// ================================================ head + mask predictor ==========================================================
namespace {
struct HeadSave {
    char *m1, *m2, *sd;           // MaskPredictor hidden activations; the dropped-out slots (fc_dropout)
    int64_t bytes;
    HeadSave(void* base, int R, int D, int h1, int h2, int dtype) {
        const int64_t es = esize(dtype);
        int64_t off = 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(base);          // (integer arithmetic: the size queries lay the arena out at address 0)
        auto take = [&](int64_t n) { char* r = reinterpret_cast<char*>(p + (uintptr_t)off); off += al256(n); return r; };
        m1 = take((int64_t)R * h1 * es); m2 = take((int64_t)R * h2 * es); sd = take((int64_t)R * D * es);
        bytes = off;
    }
};
int64_t head_tmp_bytes(int R, int D, int h1, int h2, int G, int dtype) {       // dp3 [R,G], dp2 [R,h2], dp1 [R,h1], ds_m [R,D], d(dropped slots) [R,D]
    const int64_t es = esize(dtype);
    return al256((int64_t)R * G * es) + al256((int64_t)R * h2 * es) + al256((int64_t)R * h1 * es) + 2 * al256((int64_t)R * D * es);
}
int head_check(const devias_head_args* a, const char* who) {
    DEVIAS_REQUIRE(a && a->R > 0 && a->D > 0 && a->C > 0 && a->h1 > 0 && a->h2 > 0 && a->G > 0, "%s: bad dims", who);
    DEVIAS_REQUIRE(a->dtype == DEVIAS_BF16 || a->dtype == DEVIAS_F32, "%s: bad dtype %d", who, a->dtype);
    DEVIAS_REQUIRE(a->Wh && a->bh && a->W0 && a->b0 && a->W2 && a->b2 && a->W4 && a->b4 && a->ws, "%s: null parameter / workspace", who);
    DEVIAS_REQUIRE(a->ws_bytes >= devias_head_workspace_bytes(a->R, a->D, a->C, a->h1, a->h2, a->G, a->dtype), "%s: workspace too small", who);
    return DEVIAS_OK;
}
}  // namespace

extern "C" int64_t devias_head_workspace_bytes(int32_t R, int32_t D, int32_t C, int32_t h1, int32_t h2, int32_t G, int32_t dtype) {
    int64_t w = 0;
    const int dims[][3] = {{R, C, D}, {R, h1, D}, {R, h2, h1}, {R, G, h2}, {R, h2, G}, {R, h1, h2}, {R, D, h1}, {R, D, C}};
    for (auto& d : dims) w = max64(w, gemm_ws(d[0], d[1], d[2], 0));
    const int wg[][2] = {{G, h2}, {h2, h1}, {h1, D}, {C, D}};
    for (auto& d : wg) w = max64(w, wgrad_ws(d[0], d[1], R, dtype));
    const int cs[] = {G, h2, h1, C};
    for (int n : cs) w = max64(w, devias_colsum_workspace_bytes(R, n));
    return al256(w) + 256 + head_tmp_bytes(R, D, h1, h2, G, dtype);            // backward temporaries live behind the kernel workspace
}
extern "C" int64_t devias_head_save_bytes(int32_t R, int32_t D, int32_t h1, int32_t h2, int32_t dtype) { return HeadSave(nullptr, R, D, h1, h2, dtype).bytes; }

// slots [R, D] -> Z = head(slots) [R, C], Mk = MaskPredictor(slots) [R, G] (also the last tensor of `save`)
extern "C" int devias_head_fwd(const devias_head_args* a, const void* slots, void* Z, void* Mk, void* save, void* stream) {
    RUN(head_check(a, "devias_head_fwd"));
    DEVIAS_REQUIRE(slots && Z && Mk && save, "devias_head_fwd: null pointer");
    const Ctx c{a->dtype, a->ws, a->ws_bytes, nullptr, 0, stream};
    const HeadSave s(save, a->R, a->D, a->h1, a->h2, a->dtype);
    devias_range r("head_fwd");
    const void* hin = slots;
    if (a->drop_mask) { RUN(devias_mul_mask(slots, a->drop_mask, nullptr, s.sd, a->dtype, (int64_t)a->R * a->D, stream)); hin = s.sd; }      // fc_dropout: head input only
    { Epi e; e.bias = a->bh; RUN(gemm(c, hin, a->Wh, Z, a->R, a->C, a->D, a->D, a->D, 0, 0, e)); }
    { Epi e; e.bias = a->b0; e.act = DEVIAS_ACT_RELU; RUN(gemm(c, slots, a->W0, s.m1, a->R, a->h1, a->D, a->D, a->D, 0, 0, e)); }
    { Epi e; e.bias = a->b2; e.act = DEVIAS_ACT_RELU; RUN(gemm(c, s.m1, a->W2, s.m2, a->R, a->h2, a->h1, a->h1, a->h1, 0, 0, e)); }
    { Epi e; e.bias = a->b4; e.act = DEVIAS_ACT_SIGMOID; RUN(gemm(c, s.m2, a->W4, Mk, a->R, a->G, a->h2, a->h2, a->h2, 0, 0, e)); }
    return DEVIAS_OK;
}

extern "C" int devias_head_bwd(const devias_head_args* a, const void* slots, const void* Mk, const void* save, const void* dZ, const void* dM, void* dslots,
                               const devias_head_grads* g, void* stream) {
    RUN(head_check(a, "devias_head_bwd"));
    DEVIAS_REQUIRE(slots && Mk && save && dZ && dM && dslots && g, "devias_head_bwd: null pointer");
    DEVIAS_REQUIRE(g->dWh && g->dbh && g->dW0 && g->db0 && g->dW2 && g->db2 && g->dW4 && g->db4, "devias_head_bwd: null gradient destination");
    const int R = a->R, D = a->D, C = a->C, h1 = a->h1, h2 = a->h2, G = a->G;
    const int64_t es = esize(a->dtype);
    const int64_t wk = devias_head_workspace_bytes(R, D, C, h1, h2, G, a->dtype) - head_tmp_bytes(R, D, h1, h2, G, a->dtype);
    const Ctx c{a->dtype, a->ws, wk, nullptr, 0, stream};
    char* tp = reinterpret_cast<char*>(a->ws) + wk;
    char* dp3 = tp; tp += al256((int64_t)R * G * es);
    char* dp2 = tp; tp += al256((int64_t)R * h2 * es);
    char* dp1 = tp; tp += al256((int64_t)R * h1 * es);
    char* dsm = tp; tp += al256((int64_t)R * D * es);
    char* dsd = tp;
    const HeadSave s(const_cast<void*>(save), R, D, h1, h2, a->dtype);
    devias_range r("head_bwd");
    RUN(devias_act_bwd(dM, Mk, dp3, DEVIAS_ACT_SIGMOID, a->dtype, (int64_t)R * G, stream));
    RUN(wgrad(c, dp3, s.m2, g->dW4, R, G, h2)); RUN(colsum(c, dp3, R, G, g->db4));
    { Epi e; e.act = DEVIAS_ACT_DRELU; e.aux_in = s.m2; RUN(gemm(c, dp3, a->W4, dp2, R, h2, G, G, h2, 0, 1, e)); }
    RUN(wgrad(c, dp2, s.m1, g->dW2, R, h2, h1)); RUN(colsum(c, dp2, R, h2, g->db2));
    { Epi e; e.act = DEVIAS_ACT_DRELU; e.aux_in = s.m1; RUN(gemm(c, dp2, a->W2, dp1, R, h1, h2, h2, h1, 0, 1, e)); }
    RUN(wgrad(c, dp1, slots, g->dW0, R, h1, D)); RUN(colsum(c, dp1, R, h1, g->db0));
    { Epi e; RUN(gemm(c, dp1, a->W0, dsm, R, D, h1, h1, D, 0, 1, e)); }
    if (a->drop_mask) {                  // dslots = (dZ Wh) * mask + the MaskPredictor's gradient; the head saw the dropped slots
        { Epi e; RUN(gemm(c, dZ, a->Wh, dsd, R, D, C, C, D, 0, 1, e)); }
        RUN(devias_mul_mask(dsd, a->drop_mask, dsm, dslots, a->dtype, (int64_t)R * D, stream));
        RUN(wgrad(c, dZ, s.sd, g->dWh, R, C, D));
    } else {
        { Epi e; e.res = dsm; RUN(gemm(c, dZ, a->Wh, dslots, R, D, C, C, D, 0, 1, e)); }
        RUN(wgrad(c, dZ, slots, g->dWh, R, C, D));
    }
    RUN(colsum(c, dZ, R, C, g->dbh));
    return DEVIAS_OK;
}

// ================================================ aggregation block (folded slot attention) ======================================
namespace {
// per weight set: context rows + statistics, composite weights; per layer: what backward re-reads; stacks over layers
struct AggSave {
    char *feats; float *m0, *r0;
    char* c[DEVIAS_AGG_MAX_DEPTH]; float *mc[DEVIAS_AGG_MAX_DEPTH], *rc[DEVIAS_AGG_MAX_DEPTH];
    char *Wqk[DEVIAS_AGG_MAX_DEPTH], *Wov[DEVIAS_AGG_MAX_DEPTH];
    struct Layer { char *xs_in, *qn, *z, *xs1, *f, *fpre, *fact; float *mq, *rq, *mf, *rf; } L[DEVIAS_AGG_MAX_DEPTH];
    char* qp_stack; float *attn_stack, *rsum_stack;
    char* xs_last; float *ml, *rl;
    int64_t qp_layer, attn_layer, rsum_layer;       // bytes of one layer of each stack
    int64_t bytes;
    AggSave(void* base, const devias_agg_args* a) {
        const int64_t es = esize(a->dtype), M = (int64_t)a->B * a->N, R = (int64_t)a->B * a->S, D = a->D, hD = (int64_t)a->heads * a->D, F = a->ff;
        const int nset = a->tied ? 1 : a->depth;
        int64_t off = 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(base);          // (integer arithmetic: the size queries lay the arena out at address 0)
        auto take = [&](int64_t n) { char* r = reinterpret_cast<char*>(p + (uintptr_t)off); off += al256(n); return r; };
        feats = take(M * D * es); m0 = (float*)take(M * 4); r0 = (float*)take(M * 4);
        for (int i = 0; i < nset; ++i) {
            c[i] = take(M * D * es); mc[i] = (float*)take(M * 4); rc[i] = (float*)take(M * 4);
            Wqk[i] = take(hD * D * es); Wov[i] = take(D * hD * es);
        }
        // per-layer tensors, one stack per field, the layers contiguous: the deferred weight gradients of a tied weight set reduce over all its
        // layers' rows in ONE product ([depth * R, .] operands)
        {
            char *xs_in = take(a->depth * R * D * es), *qn = take(a->depth * R * D * es), *z = take(a->depth * R * hD * es), *xs1 = take(a->depth * R * D * es);
            char *f = take(a->depth * R * D * es), *fpre = take(a->depth * R * F * es), *fact = take(a->depth * R * F * es);
            float *mq = (float*)take(a->depth * R * 4), *rq = (float*)take(a->depth * R * 4), *mf = (float*)take(a->depth * R * 4), *rf = (float*)take(a->depth * R * 4);
            for (int l = 0; l < a->depth; ++l) {
                Layer& y = L[l];
                y.xs_in = xs_in + l * R * D * es; y.qn = qn + l * R * D * es; y.z = z + l * R * hD * es; y.xs1 = xs1 + l * R * D * es; y.f = f + l * R * D * es;
                y.fpre = fpre + l * R * F * es; y.fact = fact + l * R * F * es;
                y.mq = mq + l * R; y.rq = rq + l * R; y.mf = mf + l * R; y.rf = rf + l * R;
            }
        }
        qp_layer = R * hD * es; attn_layer = (int64_t)a->B * a->heads * a->S * a->N * 4; rsum_layer = (int64_t)a->B * a->heads * a->S * 4;
        qp_stack = take(qp_layer * a->depth);
        attn_stack = (float*)take(attn_layer * a->depth);
        rsum_stack = (float*)take(rsum_layer * a->depth);
        xs_last = take(R * D * es); ml = (float*)take(R * 4); rl = (float*)take(R * 4);
        bytes = off;
    }
};
struct AggScratch {
    char *dz_stack; float* ds_stack;
    char *dxs_stack, *dfpre_stack, *dxs1_stack, *dqp_stack;      // gradients of every layer's output / FF pre-activation / attention residual / q', kept for the deferred weight gradients
    char *df, *dqn;                                    // B*S-row temporaries
    int64_t rd, rf, rh;                                // bytes of one layer of a [R, D] / [R, F] / [R, h*D] stack
    char *coef, *vec, *dc, *dfeats[2];                 // deferred context gradient
    float *gWqk[DEVIAS_AGG_MAX_DEPTH], *gWov[DEVIAS_AGG_MAX_DEPTH];   // fp32 gradient accumulators of the composite weights
    char *gqk_t, *gov_t;                               // their compute-dtype copies (bf16 mode)
    float *lnp_ffn[DEVIAS_AGG_MAX_DEPTH], *lnp_q[DEVIAS_AGG_MAX_DEPTH];   // per weight set: LayerNorm parameter-gradient partials of all its layers, [layer][part][3][D]
    int64_t lnp_layer;                                 // floats of one layer's partials
    int64_t dz_layer, ds_layer;
    int64_t bytes;
    AggScratch(void* base, const devias_agg_args* a) {
        const int64_t es = esize(a->dtype), M = (int64_t)a->B * a->N, R = (int64_t)a->B * a->S, D = a->D, hD = (int64_t)a->heads * a->D, F = a->ff;
        const int nset = a->tied ? 1 : a->depth;
        const int nl = a->tied ? a->depth : 1;
        const int64_t K = 2 * (int64_t)nl * a->heads * a->S, Np = (a->N + 7) / 8 * 8;
        int64_t off = 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(base);          // (integer arithmetic: the size queries lay the arena out at address 0)
        auto take = [&](int64_t n) { char* r = reinterpret_cast<char*>(p + (uintptr_t)off); off += al256(n); return r; };
        dz_layer = R * hD * es; ds_layer = (int64_t)a->B * a->heads * a->S * a->N * 4;
        dz_stack = take(dz_layer * a->depth); ds_stack = (float*)take(ds_layer * a->depth);
        rd = R * D * es; rf = R * F * es; rh = R * hD * es;
        dxs_stack = take((a->depth + 1) * rd);         // slot l + 1 = gradient of layer l's output; slot 0 = gradient of the latents' rows
        dfpre_stack = take(a->depth * rf); dxs1_stack = take(a->depth * rd); dqp_stack = take(a->depth * rh);
        df = take(rd); dqn = take(rd);
        coef = take((int64_t)a->B * K * Np * es); vec = take((int64_t)a->B * K * D * es); dc = take(M * D * es);
        dfeats[0] = take(M * D * es); dfeats[1] = take(M * D * es);
        for (int i = 0; i < nset; ++i) { gWqk[i] = (float*)take(hD * D * 4); gWov[i] = (float*)take(D * hD * 4); }
        gqk_t = take(hD * D * es); gov_t = take(D * hD * es);
        lnp_layer = devias_layernorm_bwd_parts_count((int)R) * 3 * D;
        for (int i = 0; i < nset; ++i) { lnp_ffn[i] = (float*)take(nl * lnp_layer * 4); lnp_q[i] = (float*)take(nl * lnp_layer * 4); }
        bytes = off;
    }
};
int agg_check(const devias_agg_args* a, const char* who) {
    DEVIAS_REQUIRE(a, "%s: null args", who);
    DEVIAS_REQUIRE(a->B > 0 && a->N > 0 && a->S >= 1 && a->S <= 4 && a->depth >= 1 && a->depth <= DEVIAS_AGG_MAX_DEPTH && a->heads > 0 && a->dh > 0 && a->ff > 0,
                   "%s: bad dims (S <= 4, depth <= %d)", who, DEVIAS_AGG_MAX_DEPTH);
    DEVIAS_REQUIRE(a->D == 384 || a->D == 512 || a->D == 768 || a->D == 1024, "%s: context dim must be 384, 512, 768 or 1024, got %d", who, a->D);
    DEVIAS_REQUIRE(a->dtype == DEVIAS_BF16 || a->dtype == DEVIAS_F32, "%s: bad dtype %d", who, a->dtype);
    DEVIAS_REQUIRE(a->norm_w && a->norm_b && a->latents && a->last_w && a->last_b && a->ws && a->save, "%s: null pointer", who);
    const int nset = a->tied ? 1 : a->depth;
    for (int i = 0; i < nset; ++i) {
        const devias_agg_layer_params& P = a->sets[i];
        DEVIAS_REQUIRE(P.Wq && P.Wk && P.Wv && P.Wo && P.W1 && P.W2 && P.bo && P.norm_w && P.norm_b && P.ctx_w && P.ctx_b && P.b1 && P.b2 && P.ffn_w && P.ffn_b,
                       "%s: null parameter in weight set %d", who, i);
    }
    DEVIAS_REQUIRE(a->ws_bytes >= devias_agg_block_workspace_bytes(a), "%s: workspace too small", who);
    return DEVIAS_OK;
}
}  // namespace

extern "C" int64_t devias_agg_block_save_bytes(const devias_agg_args* a) { return a ? AggSave(nullptr, a).bytes : 0; }
extern "C" int64_t devias_agg_block_scratch_bytes(const devias_agg_args* a) { return a ? AggScratch(nullptr, a).bytes : 0; }
extern "C" int64_t devias_agg_block_workspace_bytes(const devias_agg_args* a) {
    if (!a) return 0;
    const int M = a->B * a->N, R = a->B * a->S, D = a->D, hD = a->heads * a->D, F = a->ff;
    int64_t w = devias_layernorm_bwd_workspace_bytes(M, D);
    w = max64(w, devias_layernorm_bwd_workspace_bytes(R, D));
    w = max64(w, devias_slotf_workspace_bytes(a->B, a->S, a->N, a->heads, D));
    const int dims[][3] = {{R, hD, D}, {R, D, hD}, {R, F, D}, {R, D, F}};
    for (auto& d : dims) w = max64(w, gemm_ws(d[0], d[1], d[2], 0));
    const int RL = R * (a->tied ? a->depth : 1);
    const int wg[][2] = {{D, F}, {F, D}, {D, hD}, {hD, D}};
    for (auto& d : wg) w = max64(w, wgrad_ws(d[0], d[1], RL, a->dtype));
    w = max64(w, max64(devias_colsum_workspace_bytes(RL, D), devias_colsum_workspace_bytes(RL, F)));
    return al256(w) + 256;
}

// composite weights of one set: Wqk [h*D, D] (Wqk_h = Wk_h^T Wq_h), Wov [D, h*D] (Wov_h = Wo_h Wv_h)
static int agg_composites(const Ctx& c, const devias_agg_layer_params& P, int heads, int dh, int D, void* Wqk, void* Wov) {
    const int inner = heads * dh;
    RUN(gemm_batched(c, P.Wk, P.Wq, Wqk, 0, D, D, dh, D, D, D, (int64_t)dh * D, (int64_t)dh * D, (int64_t)D * D, heads, 1, 1));
    RUN(gemm_batched(c, P.Wo, P.Wv, Wov, 0, D, D, dh, inner, D, heads * D, dh, (int64_t)dh * D, D, heads, 0, 1));
    return DEVIAS_OK;
}

// x [B*N, D] (encoder output) -> slots [B*S, D], attn = slot softmax of the LAST layer, fp32 [B*heads, S, N] (a view into `save`: *attn_out)
extern "C" int devias_agg_block_fwd(const devias_agg_args* a, const void* x, void* slots, float** attn_out, void* stream) {
    RUN(agg_check(a, "devias_agg_block_fwd"));
    DEVIAS_REQUIRE(x && slots, "devias_agg_block_fwd: null activation");
    const int B = a->B, N = a->N, S = a->S, D = a->D, heads = a->heads, M = B * N, R = B * S, hD = heads * D, F = a->ff;
    const float scale = 1.0f / sqrtf((float)a->dh);
    const Ctx c{a->dtype, a->ws, a->ws_bytes, nullptr, 0, stream};
    AggSave s(a->save, a);
    const int nset = a->tied ? 1 : a->depth;
    devias_range r("agg_block_fwd");
    RUN(ln_fwd(c, x, a->norm_w, a->norm_b, s.feats, s.m0, s.r0, M, D, a->eps_enc));                     // modeling_slot.py:373
    for (int i = 0; i < nset; ++i) {                                                                     // context LayerNorm + composite weights, once per weight set
        RUN(ln_fwd(c, s.feats, a->sets[i].ctx_w, a->sets[i].ctx_b, s.c[i], s.mc[i], s.rc[i], M, D, a->eps_agg));
        RUN(agg_composites(c, a->sets[i], heads, a->dh, D, s.Wqk[i], s.Wov[i]));
    }
    RUN(devias_rows_broadcast(a->latents, S, D, s.L[0].xs_in, a->dtype, R, stream));                    // agg_block.py:112-114
    for (int l = 0; l < a->depth; ++l) {
        const int si = a->tied ? 0 : l;
        const devias_agg_layer_params& P = a->sets[si];
        AggSave::Layer& y = s.L[l];
        char* qp = s.qp_stack + l * s.qp_layer;
        float* attn = reinterpret_cast<float*>(reinterpret_cast<char*>(s.attn_stack) + l * s.attn_layer);
        float* rsum = reinterpret_cast<float*>(reinterpret_cast<char*>(s.rsum_stack) + l * s.rsum_layer);
        char* xs_next = l + 1 < a->depth ? s.L[l + 1].xs_in : s.xs_last;
        RUN(ln_fwd(c, y.xs_in, P.norm_w, P.norm_b, y.qn, y.mq, y.rq, R, D, a->eps_agg));
        { Epi e; RUN(gemm(c, y.qn, s.Wqk[si], qp, R, hD, D, D, D, 0, 0, e)); }
        RUN(devias_slotf_fwd(qp, s.c[si], attn, rsum, y.z, B, S, N, heads, D, scale, a->dtype, a->ws, stream));
        { Epi e; e.bias = P.bo; e.res = y.xs_in; RUN(gemm(c, y.z, s.Wov[si], y.xs1, R, D, hD, hD, hD, 0, 0, e)); }
        RUN(ln_fwd(c, y.xs1, P.ffn_w, P.ffn_b, y.f, y.mf, y.rf, R, D, a->eps_agg));
        { Epi e; e.bias = P.b1; e.act = DEVIAS_ACT_GELU; e.aux_out = y.fpre; RUN(gemm(c, y.f, P.W1, y.fact, R, F, D, D, D, 0, 0, e)); }
        { Epi e; e.bias = P.b2; e.res = y.xs1; RUN(gemm(c, y.fact, P.W2, xs_next, R, D, F, F, F, 0, 0, e)); }
    }
    RUN(ln_fwd(c, s.xs_last, a->last_w, a->last_b, slots, s.ml, s.rl, R, D, a->eps_agg));
    if (attn_out) *attn_out = reinterpret_cast<float*>(reinterpret_cast<char*>(s.attn_stack) + (a->depth - 1) * s.attn_layer);
    return DEVIAS_OK;
}

// dslots [B*S, D], dattn (optional fp32 [B*heads, S, N]: gradient arriving on the returned attention) -> dx [B*N, D] + every parameter gradient.
// Gradients of a weight set are accumulated over the layers that share it (beta = 1 on every use after the first), fixed order.
extern "C" int devias_agg_block_bwd(const devias_agg_args* a, const void* x, const void* dslots, const float* dattn, void* dx, const devias_agg_grads* g,
                                    void* scratch, int64_t scratch_bytes, void* stream) {
    RUN(agg_check(a, "devias_agg_block_bwd"));
    DEVIAS_REQUIRE(x && dslots && dx && g && scratch && aligned16(scratch), "devias_agg_block_bwd: null / unaligned argument");
    DEVIAS_REQUIRE(scratch_bytes >= devias_agg_block_scratch_bytes(a), "devias_agg_block_bwd: scratch too small");
    DEVIAS_REQUIRE(g->dnorm_w && g->dnorm_b && g->dlatents && g->dlast_w && g->dlast_b && g->dx_colsum, "devias_agg_block_bwd: null gradient destination");
    const int B = a->B, N = a->N, S = a->S, D = a->D, heads = a->heads, dh = a->dh, M = B * N, R = B * S, hD = heads * D, F = a->ff, inner = heads * dh;
    const float scale = 1.0f / sqrtf((float)dh);
    const Ctx c{a->dtype, a->ws, a->ws_bytes, nullptr, 0, stream};
    AggSave s(a->save, a);
    AggScratch t(scratch, a);
    const int nset = a->tied ? 1 : a->depth;
    for (int i = 0; i < nset; ++i) {
        const devias_agg_layer_grads& G = g->sets[i];
        DEVIAS_REQUIRE(G.dWq && G.dWk && G.dWv && G.dWo && G.dbo && G.dnorm_w && G.dnorm_b && G.dctx_w && G.dctx_b && G.dW1 && G.db1 && G.dW2 && G.db2 && G.dffn_w && G.dffn_b,
                       "devias_agg_block_bwd: null gradient destination in weight set %d", i);
    }
    devias_range r("agg_block_bwd");
    bool seen[DEVIAS_AGG_MAX_DEPTH];
    for (int i = 0; i < DEVIAS_AGG_MAX_DEPTH; ++i) seen[i] = false;
    auto dxs_of = [&](int l) { return t.dxs_stack + (int64_t)(l + 1) * t.rd; };      // gradient of layer l's output (l = -1: of the first layer's input)
    RUN(ln_bwd(c, dslots, s.xs_last, a->last_w, s.ml, s.rl, nullptr, dxs_of(a->depth - 1), g->dlast_w, g->dlast_b, 0.f, nullptr, R, D));
    for (int l = a->depth - 1; l >= 0; --l) {
        const int si = a->tied ? 0 : l;
        const devias_agg_layer_params& P = a->sets[si];
        const devias_agg_layer_grads& G = g->sets[si];
        AggSave::Layer& y = s.L[l];

        float* attn = reinterpret_cast<float*>(reinterpret_cast<char*>(s.attn_stack) + l * s.attn_layer);
        float* rsum = reinterpret_cast<float*>(reinterpret_cast<char*>(s.rsum_stack) + l * s.rsum_layer);
        char* dz = t.dz_stack + l * t.dz_layer;
        float* ds = reinterpret_cast<float*>(reinterpret_cast<char*>(t.ds_stack) + l * t.ds_layer);
        char *dxs = dxs_of(l), *dfpre = t.dfpre_stack + l * t.rf, *dxs1 = t.dxs1_stack + l * t.rd, *dqp = t.dqp_stack + l * t.rh;
        // the activation-gradient chain only: the weight (and bias) gradients of W2, W1, Wov, Wqk wait until every layer of the weight set has run
        // feed-forward: xs2 = xs1 + W2 gelu(W1 LN(xs1) + b1) + b2
        { Epi e; e.act = DEVIAS_ACT_DGELU; e.aux_in = y.fpre; RUN(gemm(c, dxs, P.W2, dfpre, R, F, D, D, F, 0, 1, e)); }
        { Epi e; RUN(gemm(c, dfpre, P.W1, t.df, R, D, F, F, D, 0, 1, e)); }
        const int li = a->tied ? l : 0;                 // this layer's place among the layers of its weight set
        RUN(devias_layernorm_bwd_parts(t.df, y.xs1, P.ffn_w, y.mf, y.rf, dxs, dxs1, t.lnp_ffn[si] + li * t.lnp_layer, R, D, a->dtype, (hipStream_t)stream));
        // cross attention: xs1 = xs + Wov z + bo
        { Epi e; RUN(gemm(c, dxs1, s.Wov[si], dz, R, hD, D, D, hD, 0, 1, e)); }
        RUN(devias_slotf_bwd(s.c[si], attn, rsum, y.z, dz, l == a->depth - 1 ? dattn : nullptr, dqp, ds, B, S, N, heads, D, scale, a->dtype, a->ws, stream));
        { Epi e; RUN(gemm(c, dqp, s.Wqk[si], t.dqn, R, D, hD, hD, D, 0, 1, e)); }
        RUN(devias_layernorm_bwd_parts(t.dqn, y.xs_in, P.norm_w, y.mq, y.rq, dxs1, dxs_of(l - 1), t.lnp_q[si] + li * t.lnp_layer, R, D, a->dtype, (hipStream_t)stream));
        seen[si] = true;
    }
    {   // the two LayerNorms' parameter gradients of every weight set: ONE fixed-order reduce per parameter over the partials of all the set's layers (four
        // reduces per set, sixteen jobs per launch) instead of a single-workgroup LayerNorm backward per layer that accumulates into them
        DeviasDeferList dl; dl.n = 0;
        const int nl = a->tied ? a->depth : 1, np = (int)devias_layernorm_bwd_parts_count(R) * nl;
        for (int si = 0; si < nset; ++si) {
            const devias_agg_layer_grads& G = g->sets[si];
            const DeviasReduceJob j[4] = {{t.lnp_ffn[si], np, 3 * D, D, G.dffn_w, 0.f}, {t.lnp_ffn[si] + D, np, 3 * D, D, G.dffn_b, 0.f},
                                          {t.lnp_q[si], np, 3 * D, D, G.dnorm_w, 0.f}, {t.lnp_q[si] + D, np, 3 * D, D, G.dnorm_b, 0.f}};
            for (int k = 0; k < 4; ++k) dl.jobs[dl.n++] = j[k];
            if (dl.n + 4 > DeviasDeferList::MAX) RUN(devias_flush_deferred(&dl, (hipStream_t)stream));
        }
        RUN(devias_flush_deferred(&dl, (hipStream_t)stream));
    }
    const char* dxs = dxs_of(-1);
    RUN(devias_rows_reduce_mod(dxs, a->dtype, R, D, S, g->dlatents, stream));
    // deferred context gradient: one pass per distinct context over all the layers that used it; composite -> parameter gradients
    const void* dfeats = nullptr;
    const int Np = (N + 7) / 8 * 8;
    for (int si = 0; si < nset; ++si) {
        const devias_agg_layer_params& P = a->sets[si];
        const devias_agg_layer_grads& G = g->sets[si];
        const int l0 = a->tied ? 0 : si, nl = a->tied ? a->depth : 1;
        const int K = 2 * nl * heads * S;
        {   // weight and bias gradients of the set's W2 / W1 / Wov / Wqk over the rows of ALL its layers (layer-major stacks): one product each instead of one per layer
            const int RL = nl * R;
            const AggSave::Layer& y0 = s.L[l0];
            const char *dxs_s = dxs_of(l0), *dfpre_s = t.dfpre_stack + l0 * t.rf, *dxs1_s = t.dxs1_stack + l0 * t.rd, *dqp_s = t.dqp_stack + l0 * t.rh;
            RUN(wgrad(c, dxs_s, y0.fact, G.dW2, RL, D, F)); RUN(colsum(c, dxs_s, RL, D, G.db2));
            RUN(wgrad(c, dfpre_s, y0.f, G.dW1, RL, F, D)); RUN(colsum(c, dfpre_s, RL, F, G.db1));
            RUN(wgrad(c, dxs1_s, y0.z, t.gWov[si], RL, D, hD)); RUN(colsum(c, dxs1_s, RL, D, G.dbo));
            RUN(wgrad(c, dqp_s, y0.qn, t.gWqk[si], RL, hD, D));
        }
        RUN(devias_slotf_pack(reinterpret_cast<const float*>(reinterpret_cast<const char*>(s.attn_stack) + l0 * s.attn_layer),
                              reinterpret_cast<const float*>(reinterpret_cast<const char*>(s.rsum_stack) + l0 * s.rsum_layer),
                              reinterpret_cast<const float*>(reinterpret_cast<const char*>(t.ds_stack) + l0 * t.ds_layer), t.dz_stack + l0 * t.dz_layer,
                              s.qp_stack + l0 * s.qp_layer, t.coef, t.vec, nl, B, S, N, Np, heads, D, scale, a->dtype, stream));
        RUN(gemm_batched(c, t.coef, t.vec, t.dc, 0, N, D, K, Np, D, D, (int64_t)K * Np, (int64_t)K * D, (int64_t)N * D, B, 1, 1));
        // gradients of to_q / to_k / to_v / to_out.weight (fp32) from those of the composites (fp32 accumulators over the layers)
        const void* gqk = t.gWqk[si];
        const void* gov = t.gWov[si];
        if (a->dtype != DEVIAS_F32) {
            RUN(devias_cast(t.gWqk[si], DEVIAS_F32, t.gqk_t, a->dtype, (int64_t)hD * D, stream)); gqk = t.gqk_t;
            RUN(devias_cast(t.gWov[si], DEVIAS_F32, t.gov_t, a->dtype, (int64_t)D * hD, stream)); gov = t.gov_t;
        }
        RUN(gemm_batched(c, P.Wk, gqk, G.dWq, 1, dh, D, D, D, D, D, (int64_t)dh * D, (int64_t)D * D, (int64_t)dh * D, heads, 0, 1));
        RUN(gemm_batched(c, P.Wq, gqk, G.dWk, 1, dh, D, D, D, D, D, (int64_t)dh * D, (int64_t)D * D, (int64_t)dh * D, heads, 0, 0));
        RUN(gemm_batched(c, gov, P.Wv, G.dWo, 1, D, dh, D, heads * D, D, inner, D, (int64_t)dh * D, dh, heads, 0, 0));
        RUN(gemm_batched(c, P.Wo, gov, G.dWv, 1, dh, D, D, inner, heads * D, D, dh, D, (int64_t)dh * D, heads, 1, 1));
        char* out = t.dfeats[si & 1];
        RUN(ln_bwd(c, t.dc, s.feats, P.ctx_w, s.mc[si], s.rc[si], dfeats, out, G.dctx_w, G.dctx_b, 0.f, nullptr, M, D));
        dfeats = out;
    }
    RUN(ln_bwd(c, dfeats, x, a->norm_w, s.m0, s.r0, nullptr, dx, g->dnorm_w, g->dnorm_b, 0.f, g->dx_colsum, M, D));
    return DEVIAS_OK;
}

// the split policies, exported so that the Python host's copies can be checked against them without a GPU (tests/test_regions_cpu.py)
extern "C" int32_t devias_policy_small_m_split(int32_t M, int32_t N, int32_t K, int32_t trans_a) { return small_m_split(M, N, K, trans_a); }
extern "C" int32_t devias_policy_wgrad_split(int32_t Nout, int32_t Kin, int32_t Mrows, int32_t dtype) { return wgrad_split(Nout, Kin, Mrows, dtype); }
