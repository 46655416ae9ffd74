// Slot cross-attention core for gfx950 (agg_block/attention.py:128-140 of the reference):
//   sim[i,j] = scale * q_i . k_j ;  A[i,j] = softmax over the SLOT axis i ;  Abar = A / (sum_j A + 1e-7) ;  o_i = sum_j Abar[i,j] v_j
// S (slots) is 2..8 while N (tokens) is ~1568 and dh = 512: the problem is a reduction over tokens, not a GEMM.
// It is HBM-bound on the K/V stream (B*N*2*h*dh elements per layer), so the kernels are organised around one
// coalesced pass over K and V per layer:
//   * a wave owns one token at a time; each lane owns 8 consecutive head dims (dh = 512 = 64 lanes x 8 -> one 16-byte
//     bf16 load per lane per token per tensor), dot products finish with wave shuffles;
//   * per-workgroup partial sums (row sums, un-normalised outputs, dq) go to a workspace and a small "finish" kernel
//     reduces them in a fixed order (deterministic; no float atomics);
//   * the K/V gradients of ALL weight-tied layers are produced by ONE pass at the end (devias_slot_attn_kv_grad) from
//     the tiny per-layer ds / A / q / dO tensors, instead of a read-modify-write of the [B,N,2,h*dh] buffer per layer.
#include "common.h"
#include <stdlib.h>

namespace {

enum { MAXS_LIMIT = 8, EPL = 8, DH = 512, TCH = 64 };   // tokens per workgroup chunk

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 x = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
    *reinterpret_cast<bf16x8*>(p) = x;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// ---------------------------------------------------------------------------------------------------------
// forward pass over K/V: writes A, and per-chunk partials  ws_r[bh][chunk][S], ws_o[bh][chunk][S][DH]
// ---------------------------------------------------------------------------------------------------------
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void slot_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                       float* __restrict__ attn, float* __restrict__ ws_r,
                                                       float* __restrict__ ws_o, int S, int N, int h, float scale) {
    __shared__ __attribute__((aligned(16))) float sm_o[3][MAXS][DH];
    __shared__ float sm_r[3][MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / h, hh = bh % h, chunk = blockIdx.x, nchunks = gridDim.x;
    const int inner = h * DH;
    float qv[MAXS][8], oacc[MAXS][8], rs[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        if (i < S) load8<T>(q + ((int64_t)b * S + i) * inner + hh * DH + lane * 8, qv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (i >= S) qv[i][e] = 0.f; oacc[i][e] = 0.f; }
        rs[i] = 0.f;
    }
    const int j0 = chunk * TCH, j1 = min(N, j0 + TCH);
    // two tokens per wave iteration (4 independent 16-byte loads in flight per lane)
    for (int jj = j0 + wave * 2; jj < j1; jj += 8) {
        const int ntok = min(2, j1 - jj);
        float k8[2][8], v8[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = min(jj + u, j1 - 1);
            const T* krow = kv + ((int64_t)b * N + j) * 2 * inner + hh * DH + lane * 8;
            load8<T>(krow, k8[u]);
            load8<T>(krow + inner, v8[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= ntok) break;
            const int j = jj + u;
            float sim[MAXS];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += qv[i][e] * k8[u][e];
                    sim[i] = wave_sum(d) * scale;
                    mx = fmaxf(mx, sim[i]);
                }
            }
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) if (i < S) { sim[i] = expf(sim[i] - mx); den += sim[i]; }
            const float inv = 1.0f / den;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    const float a = sim[i] * inv;
                    if (lane == 0) attn[((int64_t)bh * S + i) * N + j] = a;
                    rs[i] += a;
#pragma unroll
                    for (int e = 0; e < 8; ++e) oacc[i][e] += a * v8[u][e];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sm_o[wave - 1][i][lane * 8 + e] = oacc[i][e];
            if (lane == 0) sm_r[wave - 1][i] = rs[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int64_t pb = (int64_t)bh * nchunks + chunk;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
            float r = rs[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) r += sm_r[w][i];
            if (lane == 0) ws_r[pb * S + i] = r;
            float* dst = ws_o + (pb * S + i) * DH + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dst[e] = oacc[i][e] + sm_o[0][i][lane * 8 + e] + sm_o[1][i][lane * 8 + e] + sm_o[2][i][lane * 8 + e];
        }
    }
}

// finish: rsum[bh,i] = sum_chunks ws_r + 1e-7 ; o[b,i,hh*DH+d] = sum_chunks ws_o / rsum
template <typename T>
__global__ void slot_fwd_finish_kernel(const float* __restrict__ ws_r, const float* __restrict__ ws_o, float* __restrict__ rsum,
                                       T* __restrict__ o, int S, int h, int nchunks) {
    const int bh = blockIdx.x / S, i = blockIdx.x % S, b = bh / h, hh = bh % h;
    float r = 0.f;
    for (int c = 0; c < nchunks; ++c) r += ws_r[((int64_t)bh * nchunks + c) * S + i];
    r += 1e-7f;
    if (threadIdx.x == 0) rsum[(int64_t)bh * S + i] = r;
    for (int d = threadIdx.x; d < DH; d += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += ws_o[(((int64_t)bh * nchunks + c) * S + i) * DH + d];
        o[((int64_t)b * S + i) * h * DH + hh * DH + d] = from_f32<T>(s / r);
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward pass over K/V of one layer: ds[bh,i,j] and per-chunk dq partials ws_q[bh][chunk][S][DH]
// ---------------------------------------------------------------------------------------------------------
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void slot_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                       const float* __restrict__ attn, const float* __restrict__ rsum,
                                                       const T* __restrict__ o, const T* __restrict__ d_o,
                                                       const float* __restrict__ dA_ext, float* __restrict__ ds_out,
                                                       float* __restrict__ ws_q, int S, int N, int h, float scale) {
    __shared__ __attribute__((aligned(16))) float sm_q[3][MAXS][DH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / h, hh = bh % h, chunk = blockIdx.x, nchunks = gridDim.x;
    const int inner = h * DH;
    float dov[MAXS][8], dq[MAXS][8], dl[MAXS], rinv[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        dl[i] = 0.f; rinv[i] = 0.f;
        if (i < S) {
            float ov[8];
            const int64_t off = ((int64_t)b * S + i) * inner + hh * DH + lane * 8;
            load8<T>(d_o + off, dov[i]);
            load8<T>(o + off, ov);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += dov[i][e] * ov[e];
            dl[i] = wave_sum(d);
            rinv[i] = 1.0f / rsum[(int64_t)bh * S + i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (i >= S) dov[i][e] = 0.f; dq[i][e] = 0.f; }
    }
    const int j0 = chunk * TCH, j1 = min(N, j0 + TCH);
    for (int jj = j0 + wave * 2; jj < j1; jj += 8) {
        const int ntok = min(2, j1 - jj);
        float k8[2][8], v8[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = min(jj + u, j1 - 1);
            const T* krow = kv + ((int64_t)b * N + j) * 2 * inner + hh * DH + lane * 8;
            load8<T>(krow, k8[u]);
            load8<T>(krow + inner, v8[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= ntok) break;
            const int j = jj + u;
            float a[MAXS], dA[MAXS];
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += dov[i][e] * v8[u][e];
                    const float dAbar = wave_sum(d);
                    a[i] = attn[((int64_t)bh * S + i) * N + j];
                    dA[i] = (dAbar - dl[i]) * rinv[i] + (dA_ext ? dA_ext[((int64_t)bh * S + i) * N + j] : 0.f);
                    tsum += a[i] * dA[i];
                }
            }
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    const float ds = a[i] * (dA[i] - tsum);
                    if (lane == 0) ds_out[((int64_t)bh * S + i) * N + j] = ds;
                    const float w = ds * scale;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dq[i][e] += w * k8[u][e];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sm_q[wave - 1][i][lane * 8 + e] = dq[i][e];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int64_t pb = (int64_t)bh * nchunks + chunk;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
            float* dst = ws_q + (pb * S + i) * DH + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dst[e] = dq[i][e] + sm_q[0][i][lane * 8 + e] + sm_q[1][i][lane * 8 + e] + sm_q[2][i][lane * 8 + e];
        }
    }
}

template <typename T>
__global__ void slot_bwd_finish_kernel(const float* __restrict__ ws_q, T* __restrict__ dq, int S, int h, int nchunks) {
    const int bh = blockIdx.x / S, i = blockIdx.x % S, b = bh / h, hh = bh % h;
    for (int d = threadIdx.x; d < DH; d += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += ws_q[(((int64_t)bh * nchunks + c) * S + i) * DH + d];
        dq[((int64_t)b * S + i) * h * DH + hh * DH + d] = from_f32<T>(s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// deferred K/V gradient for L stacked layers that share K/V:
//   dK[j] = scale * sum_l sum_i ds_l[i,j] q_l[i] ;  dV[j] = sum_l sum_i (A_l[i,j] / rsum_l[i]) dO_l[i]
// LDS holds q_l[i] and dO_l[i] (fp32) for a group of layers (<= 64 (l,i) pairs); a wave owns one token at a time.
// ---------------------------------------------------------------------------------------------------------
enum { KVG_PAIRS = 16 };
template <typename T>
__global__ __launch_bounds__(256) void slot_kv_grad_kernel(const T* __restrict__ q_stack, const T* __restrict__ do_stack,
                                                           const float* __restrict__ ds_stack, const float* __restrict__ attn_stack,
                                                           const float* __restrict__ rsum_stack, T* __restrict__ dkv,
                                                           int L, int B, int S, int N, int h, float scale) {
    __shared__ __attribute__((aligned(16))) float sm_q[KVG_PAIRS][DH];
    __shared__ __attribute__((aligned(16))) float sm_do[KVG_PAIRS][DH];
    __shared__ float sm_rinv[KVG_PAIRS];
    __shared__ float sm_cds[KVG_PAIRS][TCH], sm_cab[KVG_PAIRS][TCH];   // per-token coefficients of this chunk (coalesced once)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / h, hh = bh % h;
    const int inner = h * DH;
    const int j0 = blockIdx.x * TCH, j1 = min(N, j0 + TCH);
    const int npairs = L * S;
    const int64_t BhSN = (int64_t)B * h * S * N;
    for (int p0 = 0; p0 < npairs; p0 += KVG_PAIRS) {
        const int np = min(KVG_PAIRS, npairs - p0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < np * (DH / 8); idx += 256) {
            const int pp = idx / (DH / 8), d8 = (idx % (DH / 8)) * 8;
            const int l = (p0 + pp) / S, i = (p0 + pp) % S;
            const int64_t off = (((int64_t)l * B + b) * S + i) * inner + hh * DH + d8;
            float t[8];
            load8<T>(q_stack + off, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) sm_q[pp][d8 + e] = t[e];
            load8<T>(do_stack + off, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) sm_do[pp][d8 + e] = t[e];
        }
        if (threadIdx.x < np) {
            const int l = (p0 + threadIdx.x) / S, i = (p0 + threadIdx.x) % S;
            sm_rinv[threadIdx.x] = 1.0f / rsum_stack[((int64_t)l * B * h + bh) * S + i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < np * TCH; idx += 256) {
            const int pp = idx / TCH, jl = idx % TCH, j = j0 + jl;
            const int l = (p0 + pp) / S, i = (p0 + pp) % S;
            float cds = 0.f, cab = 0.f;
            if (j < j1) {
                const int64_t ci = (int64_t)l * BhSN + ((int64_t)bh * S + i) * N + j;
                cds = ds_stack[ci] * scale;
                cab = attn_stack[ci] * sm_rinv[pp];
            }
            sm_cds[pp][jl] = cds; sm_cab[pp][jl] = cab;
        }
        __syncthreads();
        for (int j = j0 + wave; j < j1; j += 4) {
            T* krow = dkv + ((int64_t)b * N + j) * 2 * inner + hh * DH + lane * 8;
            float dk[8], dv[8];
            if (p0 > 0) { load8<T>(krow, dk); load8<T>(krow + inner, dv); }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
            }
            for (int pp = 0; pp < np; ++pp) {
                const float cds = sm_cds[pp][j - j0];
                const float cab = sm_cab[pp][j - j0];
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(&sm_q[pp][lane * 8]);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(&sm_q[pp][lane * 8 + 4]);
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(&sm_do[pp][lane * 8]);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(&sm_do[pp][lane * 8 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dk[e] += cds * q0[e]; dk[4 + e] += cds * q1[e];
                    dv[e] += cab * o0[e]; dv[4 + e] += cab * o1[e];
                }
            }
            store8<T>(krow, dk);
            store8<T>(krow + inner, dv);
        }
    }
}

// =========================================================================================================
// Folded form (devias_slotf_*): the K and V projections are never materialised.
//   sim[i,j] = scale * q_i . (Wk c_j) = scale * (Wk^T q_i) . c_j = scale * q'_i . c_j            q' = LN(x_s) (Wk_h^T Wq_h)^T   [B, S, h, D]
//   o_i      = sum_j Abar[i,j] (Wv c_j) = Wv (sum_j Abar[i,j] c_j) = Wv z_i                      z  = sum_j Abar[i,j] c_j       [B, S, h, D]
// (per head h; c = LayerNorm_ctx(features), D = embed dim).  Same arithmetic up to the association order of the sums; what changes is
// the stream: every layer reads the D-wide context rows c (B*N*D elements, shared by the heads) instead of the 2*h*512-wide K|V rows,
// and the [M, D] x [D, 2*h*512] projection GEMM with its dgrad and wgrad disappears -- the composite weights are D x D per head.
// Kernel structure as above: a wave owns one token at a time, each lane EPL = D / 64 consecutive context dims, per-workgroup partials
// reduced by a small finish kernel in a fixed order.  MAXS <= 4 (register budget); D in {384, 512, 768, 1024}.
// =========================================================================================================
template <typename T, int EPL> __device__ __forceinline__ void loadE(const T* p, float (&v)[EPL]);
template <int EPL> __device__ __forceinline__ void loadE_bf16(const bf16* p, float (&v)[EPL]) {
    if constexpr (EPL % 8 == 0) {
#pragma unroll
        for (int u = 0; u < EPL / 8; ++u) {
            bf16x8 x = *reinterpret_cast<const bf16x8*>(p + 8 * u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 * u + e] = (float)x[e];
        }
    } else if constexpr (EPL % 4 == 0) {
#pragma unroll
        for (int u = 0; u < EPL / 4; ++u) {
            bf16x4 x = *reinterpret_cast<const bf16x4*>(p + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * u + e] = (float)x[e];
        }
    } else {
#pragma unroll
        for (int u = 0; u < EPL / 2; ++u) {
            bf16x2 x = *reinterpret_cast<const bf16x2*>(p + 2 * u);
            v[2 * u] = (float)x[0]; v[2 * u + 1] = (float)x[1];
        }
    }
}
template <int EPL> __device__ __forceinline__ void loadE_f32(const float* p, float (&v)[EPL]) {
    if constexpr (EPL % 4 == 0) {
#pragma unroll
        for (int u = 0; u < EPL / 4; ++u) {
            f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * u);
            v[4 * u] = x[0]; v[4 * u + 1] = x[1]; v[4 * u + 2] = x[2]; v[4 * u + 3] = x[3];
        }
    } else {
#pragma unroll
        for (int u = 0; u < EPL / 2; ++u) {
            f32x2 x = *reinterpret_cast<const f32x2*>(p + 2 * u);
            v[2 * u] = x[0]; v[2 * u + 1] = x[1];
        }
    }
}
template <typename T, int EPL> struct LoadE;
template <int EPL> struct LoadE<bf16, EPL> { static __device__ __forceinline__ void run(const bf16* p, float (&v)[EPL]) { loadE_bf16<EPL>(p, v); } };
template <int EPL> struct LoadE<float, EPL> { static __device__ __forceinline__ void run(const float* p, float (&v)[EPL]) { loadE_f32<EPL>(p, v); } };

template <typename T, int MAXS, int EPL>
__global__ __launch_bounds__(256) void slotf_fwd_kernel(const T* __restrict__ qp, const T* __restrict__ ctx, float* __restrict__ attn,
                                                        float* __restrict__ ws_r, float* __restrict__ ws_z, int S, int N, int h, float scale) {
    constexpr int D = 64 * EPL;
    __shared__ __attribute__((aligned(16))) float sm_z[3][MAXS][D];
    __shared__ float sm_r[3][MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / h, hh = bh % h, chunk = blockIdx.x, nchunks = gridDim.x;
    float qv[MAXS][EPL], zacc[MAXS][EPL], rs[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        if (i < S) LoadE<T, EPL>::run(qp + ((int64_t)b * S + i) * h * D + hh * D + lane * EPL, qv[i]);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { if (i >= S) qv[i][e] = 0.f; zacc[i][e] = 0.f; }
        rs[i] = 0.f;
    }
    const int j0 = chunk * TCH, j1 = min(N, j0 + TCH);
    for (int jj = j0 + wave * 2; jj < j1; jj += 8) {               // two tokens per wave iteration
        const int ntok = min(2, j1 - jj);
        float c8[2][EPL];
#pragma unroll
        for (int u = 0; u < 2; ++u) LoadE<T, EPL>::run(ctx + ((int64_t)b * N + min(jj + u, j1 - 1)) * D + lane * EPL, c8[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= ntok) break;
            const int j = jj + u;
            float sim[MAXS];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) d += qv[i][e] * c8[u][e];
                    sim[i] = wave_sum(d) * scale;
                    mx = fmaxf(mx, sim[i]);
                }
            }
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) if (i < S) { sim[i] = expf(sim[i] - mx); den += sim[i]; }
            const float inv = 1.0f / den;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    const float a = sim[i] * inv;
                    if (lane == 0) attn[((int64_t)bh * S + i) * N + j] = a;
                    rs[i] += a;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) zacc[i][e] += a * c8[u][e];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) sm_z[wave - 1][i][lane * EPL + e] = zacc[i][e];
            if (lane == 0) sm_r[wave - 1][i] = rs[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int64_t pb = (int64_t)bh * nchunks + chunk;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
            float r = rs[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) r += sm_r[w][i];
            if (lane == 0) ws_r[pb * S + i] = r;
            float* dst = ws_z + (pb * S + i) * D + lane * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                dst[e] = zacc[i][e] + sm_z[0][i][lane * EPL + e] + sm_z[1][i][lane * EPL + e] + sm_z[2][i][lane * EPL + e];
        }
    }
}

// sum of n values `stride` floats apart, added in index order with eight loads in flight (a chain of n dependent load latencies otherwise: these finish kernels are all latency)
__device__ __forceinline__ float sum_strided8(const float* __restrict__ p, int n, int64_t stride) {
    float s = 0.f;
    int c = 0;
    for (; c + 7 < n; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < n; ++c) s += p[(int64_t)c * stride];
    return s;
}

// finish: rsum[bh,i] = sum_chunks ws_r + 1e-7 ; z[b,i,hh*D+d] = sum_chunks ws_z / rsum.  (bf16 path: launched with one thread per column -- with 256 threads a
// thread summed D / 256 columns one after the other, each a round trip to memory: 8.0 -> see DESIGN.md section 5 round 5)
template <typename T>
__global__ void slotf_fwd_finish_kernel(const float* __restrict__ ws_r, const float* __restrict__ ws_z, float* __restrict__ rsum,
                                        T* __restrict__ z, int S, int h, int D, int nchunks) {
    const int bh = blockIdx.x / S, i = blockIdx.x % S, b = bh / h, hh = bh % h;
    float r = sum_strided8(ws_r + (int64_t)bh * nchunks * S + i, nchunks, S);
    r += 1e-7f;
    if (threadIdx.x == 0) rsum[(int64_t)bh * S + i] = r;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float s = sum_strided8(ws_z + ((int64_t)bh * nchunks * S + i) * D + d, nchunks, (int64_t)S * D);
        z[((int64_t)b * S + i) * h * D + hh * D + d] = from_f32<T>(s / r);
    }
}

// backward of one layer: ds[bh,i,j] and per-chunk dq' partials.  delta_i = dz_i . z_i (= dO_i . o_i of the unfolded form)
template <typename T, int MAXS, int EPL>
__global__ __launch_bounds__(256) void slotf_bwd_kernel(const T* __restrict__ ctx, const float* __restrict__ attn, const float* __restrict__ rsum,
                                                        const T* __restrict__ z, const T* __restrict__ dz, const float* __restrict__ dA_ext,
                                                        float* __restrict__ ds_out, float* __restrict__ ws_q, int S, int N, int h, float scale) {
    constexpr int D = 64 * EPL;
    __shared__ __attribute__((aligned(16))) float sm_q[3][MAXS][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / h, hh = bh % h, chunk = blockIdx.x, nchunks = gridDim.x;
    float dzv[MAXS][EPL], dq[MAXS][EPL], dl[MAXS], rinv[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        dl[i] = 0.f; rinv[i] = 0.f;
        if (i < S) {
            float zv[EPL];
            const int64_t off = ((int64_t)b * S + i) * h * D + hh * D + lane * EPL;
            LoadE<T, EPL>::run(dz + off, dzv[i]);
            LoadE<T, EPL>::run(z + off, zv);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) d += dzv[i][e] * zv[e];
            dl[i] = wave_sum(d);
            rinv[i] = 1.0f / rsum[(int64_t)bh * S + i];
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) { if (i >= S) dzv[i][e] = 0.f; dq[i][e] = 0.f; }
    }
    const int j0 = chunk * TCH, j1 = min(N, j0 + TCH);
    for (int jj = j0 + wave * 2; jj < j1; jj += 8) {
        const int ntok = min(2, j1 - jj);
        float c8[2][EPL];
#pragma unroll
        for (int u = 0; u < 2; ++u) LoadE<T, EPL>::run(ctx + ((int64_t)b * N + min(jj + u, j1 - 1)) * D + lane * EPL, c8[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= ntok) break;
            const int j = jj + u;
            float a[MAXS], dA[MAXS];
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) d += dzv[i][e] * c8[u][e];
                    const float dAbar = wave_sum(d);
                    a[i] = attn[((int64_t)bh * S + i) * N + j];
                    dA[i] = (dAbar - dl[i]) * rinv[i] + (dA_ext ? dA_ext[((int64_t)bh * S + i) * N + j] : 0.f);
                    tsum += a[i] * dA[i];
                }
            }
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {
                if (i < S) {
                    const float ds = a[i] * (dA[i] - tsum);
                    if (lane == 0) ds_out[((int64_t)bh * S + i) * N + j] = ds;
                    const float w = ds * scale;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) dq[i][e] += w * c8[u][e];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) sm_q[wave - 1][i][lane * EPL + e] = dq[i][e];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int64_t pb = (int64_t)bh * nchunks + chunk;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) if (i < S) {
            float* dst = ws_q + (pb * S + i) * D + lane * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                dst[e] = dq[i][e] + sm_q[0][i][lane * EPL + e] + sm_q[1][i][lane * EPL + e] + sm_q[2][i][lane * EPL + e];
        }
    }
}

template <typename T>
__global__ void slotf_bwd_finish_kernel(const float* __restrict__ ws_q, T* __restrict__ dqp, int S, int h, int D, int nchunks) {
    const int bh = blockIdx.x / S, i = blockIdx.x % S, b = bh / h, hh = bh % h;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float s = sum_strided8(ws_q + ((int64_t)bh * nchunks * S + i) * D + d, nchunks, (int64_t)S * D);
        dqp[((int64_t)b * S + i) * h * D + hh * D + d] = from_f32<T>(s);
    }
}

// Operands of the deferred context gradient of L stacked layers that share the context rows:
//   dc[b, j, :] = sum over k = (l, w, hh, i) of coef[b][k][j] * vec[b][k][:]
//     w = 0: coef = A_l[i,j] / rsum_l[i]   vec = dz_l[b,i,hh,:]        (from z = sum_j Abar c_j)
//     w = 1: coef = scale * ds_l[i,j]      vec = q'_l[b,i,hh,:]        (from sim = scale q' . c_j)
// i.e. one [N, K] x [K, D] product per clip (K = 2 L h S), run as a batched GEMM by the caller.  coef rows are padded to Np tokens (zeros).
template <typename T>
__global__ __launch_bounds__(256) void slotf_pack_kernel(const float* __restrict__ attn_stack, const float* __restrict__ rsum_stack,
                                                         const float* __restrict__ ds_stack, const T* __restrict__ dz_stack,
                                                         const T* __restrict__ qp_stack, T* __restrict__ coef, T* __restrict__ vec,
                                                         int L, int B, int S, int N, int Np, int h, int D, float scale) {
    const int K = 2 * L * h * S;
    const int b = blockIdx.y, k = blockIdx.x;
    const int i = k % S, hh = (k / S) % h, w = (k / (S * h)) % 2, l = k / (2 * S * h);
    const int64_t row = ((int64_t)l * B * h + (int64_t)b * h + hh) * S + i;         // row of the [L, B*h, S, N] stacks
    T* crow = coef + ((int64_t)b * K + k) * Np;
    if (w == 0) {
        const float rinv = 1.0f / rsum_stack[row];
        for (int j = threadIdx.x; j < Np; j += 256) crow[j] = from_f32<T>(j < N ? attn_stack[row * N + j] * rinv : 0.f);
    } else {
        for (int j = threadIdx.x; j < Np; j += 256) crow[j] = from_f32<T>(j < N ? ds_stack[row * N + j] * scale : 0.f);
    }
    const T* src = (w == 0 ? dz_stack : qp_stack) + (((int64_t)l * B + b) * S + i) * h * D + (int64_t)hh * D;
    T* vrow = vec + ((int64_t)b * K + k) * D;
    for (int d = threadIdx.x; d < D; d += 256) vrow[d] = src[d];
}

// =========================================================================================================
// Folded form on the matrix cores (bf16): the two reductions of a layer are skinny GEMMs with R = h*S <= 16 slot-head rows,
//   sim^T [tokens x R] = C [tokens x D] Q'^T [D x R]        and        Z^T [D x R] = C^T [D x tokens] A^T [tokens x R]
// (backward: dAbar^T = C dZ^T and dQ'^T = C^T (scale dS)^T), so the per-token work that the kernels above do in VALU dot products and
// wave reductions (they are VALU-bound: ~120 vector instructions per token and head) becomes 24 MFMAs per 32-token tile and wave.
//   workgroup = one clip x TCHM tokens, 4 waves; wave w owns the context columns [w*D/4, (w+1)*D/4): it stages ITS column slice of the
//   32-token tile into its own LDS region (LDS-DMA, transposed-read image blocks of [32 tokens][64 cols]) and uses it for both products:
//   as the K-split of the first (partial sim, exchanged through LDS and summed by every wave in a fixed order) and as the output-row
//   slice of the second.  The accumulator tile of the first product -- lane = slot-head row, 4 tokens per register quad -- is, packed to
//   bf16, directly the B operand of the second (same k-permutation trick as the encoder attention kernels).  Softmax over the slot
//   axis = S adjacent lanes.  Partials per workgroup in the layout of the VALU kernels (same finish kernels).
// =========================================================================================================
enum { TCHM = 128 };                                   // tokens per workgroup (4 tiles of 32)
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;
typedef __attribute__((address_space(3))) bf16x4* lds_b4ptr;

__device__ __forceinline__ int sm_tr_off(int row, int col) {        // [rows][64 cols] bf16 image, 32-byte windows XOR ((row >> 1) & 3)
    return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + (col & 15) * 2;
}
// row fragment: lane holds img[row = base + (lane & 15)][32*ks + 8*(lane >> 4) .. +8]
__device__ __forceinline__ bf16x8 sm_frag_rows(const char* img, int base, int ks, int lane) {
    const int row = base + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + sm_tr_off(row, 32 * ks + 8 * (lane >> 4)));
}
// transposed fragment over the 32 rows: lane holds img[row = 16*(j >> 2) + 4g + (j & 3)][col = cbase + (lane & 15)], j = 0..7
__device__ __forceinline__ bf16x8 sm_frag_tr(const char* img, int cbase, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int r0 = 4 * g + (c >> 2), col = cbase + 4 * (c & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4ptr)(img + sm_tr_off(r0, col)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4ptr)(img + sm_tr_off(r0 + 16, col)));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
__device__ __forceinline__ f32x4 sm_mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// NB = 64-column blocks per wave (D = 256 * NB): 3 for ViT-B (768), 4 for ViT-L (1024), 2 for D = 512
// xor-1 / xor-2 lane exchange inside a quad by DPP quad_perm (no LDS crossbar round trip); o is 1 or 2
__device__ __forceinline__ float quad_xor(float x, int o) {
    return o == 1 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false))
                  : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
}
template <int NB, bool BWD>
__global__ __launch_bounds__(256) void slotm_kernel(const bf16* __restrict__ rowsrc,   // forward: q' [B*S, h*D]; backward: dz [B*S, h*D]
                                                    const bf16* __restrict__ ctx, float* __restrict__ attn,      // forward: written; backward: read
                                                    const float* __restrict__ rsum, const bf16* __restrict__ z, const float* __restrict__ dA_ext,
                                                    float* __restrict__ ds_out, float* __restrict__ ws_r, float* __restrict__ ws_z,
                                                    int S, int N, int h, float scale) {
    constexpr int D = 256 * NB, WD = 64 * NB;          // context dim, columns per wave
    __shared__ __attribute__((aligned(16))) char smem[4 * NB * 4096 + 4 * 2048 + 256];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myimg = smem + wave * NB * 4096;             // this wave's [32 tokens][WD cols] slice: NB blocks of [32][64]
    float* xch = reinterpret_cast<float*>(smem + 4 * NB * 4096);                 // [4 waves][2 subtiles][64 lanes][4] partial sim
    float* s_dl = reinterpret_cast<float*>(smem + 4 * NB * 4096 + 4 * 2048);     // [4 waves][16] partial delta (backward)
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int R = h * S;
    const int hh = c / S, si = c - hh * S;             // this lane's slot-head row: rho = c = hh * S + si (valid while c < R)
    const bool rvalid = c < R;
    // ---- this lane's row operand (q' or dz), its wave's column slice: element k of k-step (blk, ks): rows[rho][wave*WD + 64*blk + 32*ks + 8g + k] ----
    bf16x8 rowf[NB][2];
    float dl = 0.f, rinv = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int64_t off = ((int64_t)b * S + si) * h * D + (int64_t)hh * D + wave * WD + 64 * blk + 32 * ks + 8 * g;
            bf16x8 v = {};
            if (rvalid) v = *reinterpret_cast<const bf16x8*>(rowsrc + off);
            rowf[blk][ks] = v;
            if constexpr (BWD) {
                if (rvalid) {
                    const bf16x8 zz = *reinterpret_cast<const bf16x8*>(z + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl += (float)v[e] * (float)zz[e];
                }
            }
        }
    if constexpr (BWD) {                               // delta[rho] = dz[rho] . z[rho]: over this lane's k-groups, then over the waves (fixed order)
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0) s_dl[wave * 16 + c] = dl;
        __syncthreads();
        dl = s_dl[c] + s_dl[16 + c] + s_dl[32 + c] + s_dl[48 + c];
        rinv = rvalid ? 1.0f / rsum[((int64_t)b * h + hh) * S + si] : 0.f;
    }
    f32x4 zacc[NB * 4];                                // Z^T / dQ'^T slice: rows d = wave*WD + 16*dt + 4g + r, col rho = c
#pragma unroll
    for (int i = 0; i < NB * 4; ++i) zacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rs = 0.f;
    const int j0 = chunk * TCHM, j1 = min(N, j0 + TCHM);
    const bf16* cbase = ctx + (int64_t)b * N * D + wave * WD;
    for (int t0 = j0; t0 < j1; t0 += 32) {
        // ---- stage this wave's slice of the tile: NB blocks x 4 LDS-DMA instructions (8 rows x 128 B each); rows past N re-read row N-1 ----
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3), slot = lane & 7;
                const int chunk16 = (((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1);
                const bf16* src = cbase + (int64_t)min(t0 + row, N - 1) * D + 64 * blk + chunk16 * 8;
                __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(myimg + blk * 4096 + i * 1024), 16, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- partial sim^T over this wave's columns: acc[sub] = tokens 16*sub + 4g + r (rows) x rho c (col) -----------------------------------
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    acc[sub] = sm_mfma(sm_frag_rows(myimg + blk * 4096, 16 * sub, ks, lane), rowf[blk][ks], acc[sub]);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) *reinterpret_cast<f32x4*>(xch + ((wave * 2 + sub) * 64 + lane) * 4) = acc[sub];
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 t = *reinterpret_cast<const f32x4*>(xch + ((0 * 2 + sub) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(xch + ((w * 2 + sub) * 64 + lane) * 4);
            acc[sub] = t;
        }
        // ---- per token: softmax over the slot axis (S adjacent lanes) / its backward -----------------------------------------------------
        bf16x8 bop;                                    // B operand of the second product: k = token (16*(j >> 2) + 4g + (j & 3)), col = rho
        float vals[8];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int tok = t0 + 16 * sub + 4 * g;     // first of this lane's 4 tokens
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, e4 = {0.f, 0.f, 0.f, 0.f};
            const int64_t arow = (((int64_t)b * h + hh) * S + si) * N;
            if constexpr (BWD) {
                if (rvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (tok + r < j1) {
                            a4[r] = attn[arow + tok + r];
                            if (dA_ext) e4[r] = dA_ext[arow + tok + r];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                if constexpr (!BWD) {
                    float sv = rvalid ? acc[sub][r] * scale : -INFINITY;
                    float mx = sv;
                    _Pragma("unroll") for (int o = 1; o < S; o <<= 1) mx = fmaxf(mx, quad_xor(mx, o));          // S is 1, 2 or 4: lanes of one head are aligned
                    float ev = rvalid ? expf(sv - mx) : 0.f;
                    float den = ev;
                    _Pragma("unroll") for (int o = 1; o < S; o <<= 1) den += quad_xor(den, o);
                    v = (rvalid && tok + r < j1) ? ev / den : 0.f;
                    rs += v;
                } else {
                    const float dA = (acc[sub][r] - dl) * rinv + e4[r];
                    float tsum = a4[r] * dA;
                    _Pragma("unroll") for (int o = 1; o < S; o <<= 1) tsum += quad_xor(tsum, o);
                    v = (rvalid && tok + r < j1) ? a4[r] * (dA - tsum) : 0.f;
                }
                vals[4 * sub + r] = v;
            }
            // ---- attention (forward) / dS (backward) rows: one wave per subtile writes them -----------------------------------------------
            if (rvalid && wave == sub) {
                float* dst = (BWD ? ds_out : attn) + arow + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (tok + r < j1) dst[r] = vals[4 * sub + r];
            }
        }
        if constexpr (BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vals[e] *= scale;
        }
        bop = bf16x8{(bf16)vals[0], (bf16)vals[1], (bf16)vals[2], (bf16)vals[3], (bf16)vals[4], (bf16)vals[5], (bf16)vals[6], (bf16)vals[7]};
        // ---- Z^T / dQ'^T slice += C^T (this wave's columns) x bop ---------------------------------------------------------------------------
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                zacc[blk * 4 + dt] = sm_mfma(sm_frag_tr(myimg + blk * 4096, 16 * dt, lane), bop, zacc[blk * 4 + dt]);
        __syncthreads();                               // the exchange buffer is free again (the images are private to the wave)
    }
    // ---- partials of this workgroup, in the layout the finish kernels expect ---------------------------------------------------------------------
    if (rvalid) {
        const int64_t pb = ((int64_t)b * h + hh) * nchunks + chunk;
        if constexpr (!BWD) {
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            if (wave == 0 && g == 0) ws_r[pb * S + si] = rs;
        }
        float* dst = ws_z + (pb * S + si) * D + wave * WD + 4 * g;
#pragma unroll
        for (int i = 0; i < NB * 4; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = zacc[i];
    }
}

}  // namespace

extern "C" int64_t devias_slot_attn_workspace_bytes(int32_t B, int32_t S, int32_t N, int32_t h, int32_t dh) {
    int nchunks = cdiv(N, TCH);
    return (int64_t)B * h * nchunks * S * (dh + 1) * 4 + 64;
}

#define SLOT_COMMON_CHECKS(name)                                                                              \
    DEVIAS_REQUIRE(dh == DH, name ": only dim_head == 512 is built (agg_block/agg_block.py:83), got %d", dh); \
    DEVIAS_REQUIRE(S >= 1 && S <= MAXS_LIMIT, name ": 1 <= num_latents <= 8 supported, got %d", S);                \
    DEVIAS_REQUIRE(B > 0 && N > 0 && h > 0 && (int64_t)B * h <= 65535, name ": bad B/N/h");                  \
    DEVIAS_REQUIRE(dtype == DEVIAS_BF16 || dtype == DEVIAS_F32, name ": bad dtype %d", dtype)

extern "C" int devias_slot_attn_fwd(const void* q, const void* kv, float* attn, float* rsum, void* o, int32_t B, int32_t S,
                                    int32_t N, int32_t h, int32_t dh, float scale, int32_t dtype, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SLOT_COMMON_CHECKS("devias_slot_attn_fwd");
    DEVIAS_REQUIRE(q && kv && attn && rsum && o && ws, "devias_slot_attn_fwd: null pointer");
    DEVIAS_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(o) && aligned16(ws), "devias_slot_attn_fwd: unaligned pointer");
    const int nchunks = cdiv(N, TCH);
    float* ws_r = ws;
    float* ws_o = ws + (((int64_t)B * h * nchunks * S + 3) & ~(int64_t)3);
    dim3 grid(nchunks, B * h), block(256);
    if (dtype == DEVIAS_BF16) {
        if (S <= 2) hipLaunchKernelGGL((slot_fwd_kernel<bf16, 2>), grid, block, 0, st, (const bf16*)q, (const bf16*)kv, attn, ws_r, ws_o, S, N, h, scale);
        else if (S <= 4) hipLaunchKernelGGL((slot_fwd_kernel<bf16, 4>), grid, block, 0, st, (const bf16*)q, (const bf16*)kv, attn, ws_r, ws_o, S, N, h, scale);
        else hipLaunchKernelGGL((slot_fwd_kernel<bf16, 8>), grid, block, 0, st, (const bf16*)q, (const bf16*)kv, attn, ws_r, ws_o, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slot_attn_fwd");
        hipLaunchKernelGGL((slot_fwd_finish_kernel<bf16>), dim3(B * h * S), dim3(256), 0, st, ws_r, ws_o, rsum, (bf16*)o, S, h, nchunks);
    } else {
        if (S <= 2) hipLaunchKernelGGL((slot_fwd_kernel<float, 2>), grid, block, 0, st, (const float*)q, (const float*)kv, attn, ws_r, ws_o, S, N, h, scale);
        else if (S <= 4) hipLaunchKernelGGL((slot_fwd_kernel<float, 4>), grid, block, 0, st, (const float*)q, (const float*)kv, attn, ws_r, ws_o, S, N, h, scale);
        else hipLaunchKernelGGL((slot_fwd_kernel<float, 8>), grid, block, 0, st, (const float*)q, (const float*)kv, attn, ws_r, ws_o, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slot_attn_fwd");
        hipLaunchKernelGGL((slot_fwd_finish_kernel<float>), dim3(B * h * S), dim3(256), 0, st, ws_r, ws_o, rsum, (float*)o, S, h, nchunks);
    }
    DEVIAS_CHECK_LAUNCH("devias_slot_attn_fwd(finish)");
    return DEVIAS_OK;
}

extern "C" int devias_slot_attn_bwd(const void* q, const void* kv, const float* attn, const float* rsum, const void* o,
                                    const void* d_o, const float* d_attn_ext, void* dq, float* ds, int32_t B, int32_t S,
                                    int32_t N, int32_t h, int32_t dh, float scale, int32_t dtype, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SLOT_COMMON_CHECKS("devias_slot_attn_bwd");
    DEVIAS_REQUIRE(q && kv && attn && rsum && o && d_o && dq && ds && ws, "devias_slot_attn_bwd: null pointer");
    DEVIAS_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(o) && aligned16(d_o) && aligned16(dq) && aligned16(ws),
                   "devias_slot_attn_bwd: unaligned pointer");
    const int nchunks = cdiv(N, TCH);
    dim3 grid(nchunks, B * h), block(256);
    if (dtype == DEVIAS_BF16) {
#define SB_ARGS (const bf16*)q, (const bf16*)kv, attn, rsum, (const bf16*)o, (const bf16*)d_o, d_attn_ext, ds, ws, S, N, h, scale
        if (S <= 2) hipLaunchKernelGGL((slot_bwd_kernel<bf16, 2>), grid, block, 0, st, SB_ARGS);
        else if (S <= 4) hipLaunchKernelGGL((slot_bwd_kernel<bf16, 4>), grid, block, 0, st, SB_ARGS);
        else hipLaunchKernelGGL((slot_bwd_kernel<bf16, 8>), grid, block, 0, st, SB_ARGS);
#undef SB_ARGS
        DEVIAS_CHECK_LAUNCH("devias_slot_attn_bwd");
        hipLaunchKernelGGL((slot_bwd_finish_kernel<bf16>), dim3(B * h * S), dim3(256), 0, st, ws, (bf16*)dq, S, h, nchunks);
    } else {
#define SB_ARGS (const float*)q, (const float*)kv, attn, rsum, (const float*)o, (const float*)d_o, d_attn_ext, ds, ws, S, N, h, scale
        if (S <= 2) hipLaunchKernelGGL((slot_bwd_kernel<float, 2>), grid, block, 0, st, SB_ARGS);
        else if (S <= 4) hipLaunchKernelGGL((slot_bwd_kernel<float, 4>), grid, block, 0, st, SB_ARGS);
        else hipLaunchKernelGGL((slot_bwd_kernel<float, 8>), grid, block, 0, st, SB_ARGS);
#undef SB_ARGS
        DEVIAS_CHECK_LAUNCH("devias_slot_attn_bwd");
        hipLaunchKernelGGL((slot_bwd_finish_kernel<float>), dim3(B * h * S), dim3(256), 0, st, ws, (float*)dq, S, h, nchunks);
    }
    DEVIAS_CHECK_LAUNCH("devias_slot_attn_bwd(finish)");
    return DEVIAS_OK;
}

extern "C" int devias_slot_attn_kv_grad(const void* q_stack, const void* do_stack, const float* ds_stack,
                                        const float* attn_stack, const float* rsum_stack, void* dkv, int32_t L, int32_t B,
                                        int32_t S, int32_t N, int32_t h, int32_t dh, float scale, int32_t dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SLOT_COMMON_CHECKS("devias_slot_attn_kv_grad");
    DEVIAS_REQUIRE(L >= 1 && q_stack && do_stack && ds_stack && attn_stack && rsum_stack && dkv, "devias_slot_attn_kv_grad: bad args");
    DEVIAS_REQUIRE(aligned16(q_stack) && aligned16(do_stack) && aligned16(dkv), "devias_slot_attn_kv_grad: unaligned pointer");
    dim3 grid(cdiv(N, TCH), B * h), block(256);
    if (dtype == DEVIAS_BF16)
        hipLaunchKernelGGL((slot_kv_grad_kernel<bf16>), grid, block, 0, st, (const bf16*)q_stack, (const bf16*)do_stack, ds_stack,
                           attn_stack, rsum_stack, (bf16*)dkv, L, B, S, N, h, scale);
    else
        hipLaunchKernelGGL((slot_kv_grad_kernel<float>), grid, block, 0, st, (const float*)q_stack, (const float*)do_stack, ds_stack,
                           attn_stack, rsum_stack, (float*)dkv, L, B, S, N, h, scale);
    DEVIAS_CHECK_LAUNCH("devias_slot_attn_kv_grad");
    return DEVIAS_OK;
}


// ---- folded form ----------------------------------------------------------------------------------------------------------------------------
extern "C" int64_t devias_slotf_workspace_bytes(int32_t B, int32_t S, int32_t N, int32_t h, int32_t D) {
    int nchunks = cdiv(N, TCH);
    return (int64_t)B * h * nchunks * S * (D + 1) * 4 + 64;
}

// the matrix-core kernels: bf16, h * S <= 16 slot-head rows, S a power of two (softmax across S aligned adjacent lanes), D = 256 * {2, 3, 4}
static bool slotm_ok(int B, int S, int h, int D, int dtype) {
    static const int on = [] { const char* e = getenv("DEVIAS_SLOT_MFMA"); return e ? atoi(e) : 1; }();
    return on && dtype == DEVIAS_BF16 && h * S <= 16 && (S == 1 || S == 2 || S == 4) && (D == 512 || D == 768 || D == 1024) && B <= 65535;
}

#define SLOTF_CHECKS(name)                                                                                                  \
    DEVIAS_REQUIRE(D == 384 || D == 512 || D == 768 || D == 1024, name ": context dim must be 384, 512, 768 or 1024, got %d", D); \
    DEVIAS_REQUIRE(S >= 1 && S <= 4, name ": 1 <= num_latents <= 4 in the folded form, got %d", S);                          \
    DEVIAS_REQUIRE(B > 0 && N > 0 && h > 0 && (int64_t)B * h <= 65535, name ": bad B/N/h");                                 \
    DEVIAS_REQUIRE(dtype == DEVIAS_BF16 || dtype == DEVIAS_F32, name ": bad dtype %d", dtype)

#define SLOTF_DISPATCH(KERNEL, TT, ...)                                                                  \
    do {                                                                                                 \
        if (S <= 2) {                                                                                    \
            if (D == 384) hipLaunchKernelGGL((KERNEL<TT, 2, 6>), grid, block, 0, st, __VA_ARGS__);       \
            else if (D == 512) hipLaunchKernelGGL((KERNEL<TT, 2, 8>), grid, block, 0, st, __VA_ARGS__);  \
            else if (D == 768) hipLaunchKernelGGL((KERNEL<TT, 2, 12>), grid, block, 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<TT, 2, 16>), grid, block, 0, st, __VA_ARGS__);               \
        } else {                                                                                         \
            if (D == 384) hipLaunchKernelGGL((KERNEL<TT, 4, 6>), grid, block, 0, st, __VA_ARGS__);       \
            else if (D == 512) hipLaunchKernelGGL((KERNEL<TT, 4, 8>), grid, block, 0, st, __VA_ARGS__);  \
            else if (D == 768) hipLaunchKernelGGL((KERNEL<TT, 4, 12>), grid, block, 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<TT, 4, 16>), grid, block, 0, st, __VA_ARGS__);               \
        }                                                                                                \
    } while (0)

extern "C" int devias_slotf_fwd(const void* qp, const void* ctx, float* attn, float* rsum, void* z, int32_t B, int32_t S, int32_t N,
                                int32_t h, int32_t D, float scale, int32_t dtype, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SLOTF_CHECKS("devias_slotf_fwd");
    DEVIAS_REQUIRE(qp && ctx && attn && rsum && z && ws, "devias_slotf_fwd: null pointer");
    DEVIAS_REQUIRE(aligned16(qp) && aligned16(ctx) && aligned16(z) && aligned16(ws), "devias_slotf_fwd: unaligned pointer");
    const bool mfma = slotm_ok(B, S, h, D, dtype);
    const int nchunks = mfma ? cdiv(N, TCHM) : cdiv(N, TCH);
    float* ws_r = ws;
    float* ws_z = ws + (((int64_t)B * h * nchunks * S + 3) & ~(int64_t)3);
    if (mfma) {
        dim3 gm(nchunks, B);
#define SLOTM_F(NB_) hipLaunchKernelGGL((slotm_kernel<NB_, false>), gm, dim3(256), 0, st, (const bf16*)qp, (const bf16*)ctx, attn, (const float*)nullptr, \
                                          (const bf16*)nullptr, (const float*)nullptr, (float*)nullptr, ws_r, ws_z, S, N, h, scale)
        if (D == 512) SLOTM_F(2); else if (D == 768) SLOTM_F(3); else SLOTM_F(4);
#undef SLOTM_F
        DEVIAS_CHECK_LAUNCH("devias_slotf_fwd(mfma)");
        hipLaunchKernelGGL((slotf_fwd_finish_kernel<bf16>), dim3(B * h * S), dim3(D <= 1024 ? (D + 63) / 64 * 64 : 256), 0, st, ws_r, ws_z, rsum, (bf16*)z, S, h, D, nchunks);
        DEVIAS_CHECK_LAUNCH("devias_slotf_fwd(finish)");
        return DEVIAS_OK;
    }
    dim3 grid(nchunks, B * h), block(256);
    if (dtype == DEVIAS_BF16) {
        SLOTF_DISPATCH(slotf_fwd_kernel, bf16, (const bf16*)qp, (const bf16*)ctx, attn, ws_r, ws_z, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slotf_fwd");
        hipLaunchKernelGGL((slotf_fwd_finish_kernel<bf16>), dim3(B * h * S), dim3(D <= 1024 ? (D + 63) / 64 * 64 : 256), 0, st, ws_r, ws_z, rsum, (bf16*)z, S, h, D, nchunks);
    } else {
        SLOTF_DISPATCH(slotf_fwd_kernel, float, (const float*)qp, (const float*)ctx, attn, ws_r, ws_z, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slotf_fwd");
        hipLaunchKernelGGL((slotf_fwd_finish_kernel<float>), dim3(B * h * S), dim3(256), 0, st, ws_r, ws_z, rsum, (float*)z, S, h, D, nchunks);
    }
    DEVIAS_CHECK_LAUNCH("devias_slotf_fwd(finish)");
    return DEVIAS_OK;
}

extern "C" int devias_slotf_bwd(const void* ctx, const float* attn, const float* rsum, const void* z, const void* dz,
                                const float* d_attn_ext, void* dqp, float* ds, int32_t B, int32_t S, int32_t N, int32_t h, int32_t D,
                                float scale, int32_t dtype, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SLOTF_CHECKS("devias_slotf_bwd");
    DEVIAS_REQUIRE(ctx && attn && rsum && z && dz && dqp && ds && ws, "devias_slotf_bwd: null pointer");
    DEVIAS_REQUIRE(aligned16(ctx) && aligned16(z) && aligned16(dz) && aligned16(dqp) && aligned16(ws), "devias_slotf_bwd: unaligned pointer");
    const bool mfma = slotm_ok(B, S, h, D, dtype);
    const int nchunks = mfma ? cdiv(N, TCHM) : cdiv(N, TCH);
    if (mfma) {
        dim3 gm(nchunks, B);
#define SLOTM_B(NB_) hipLaunchKernelGGL((slotm_kernel<NB_, true>), gm, dim3(256), 0, st, (const bf16*)dz, (const bf16*)ctx, const_cast<float*>(attn), rsum, \
                                          (const bf16*)z, d_attn_ext, ds, (float*)nullptr, ws, S, N, h, scale)
        if (D == 512) SLOTM_B(2); else if (D == 768) SLOTM_B(3); else SLOTM_B(4);
#undef SLOTM_B
        DEVIAS_CHECK_LAUNCH("devias_slotf_bwd(mfma)");
        hipLaunchKernelGGL((slotf_bwd_finish_kernel<bf16>), dim3(B * h * S), dim3(D <= 1024 ? (D + 63) / 64 * 64 : 256), 0, st, ws, (bf16*)dqp, S, h, D, nchunks);
        DEVIAS_CHECK_LAUNCH("devias_slotf_bwd(finish)");
        return DEVIAS_OK;
    }
    dim3 grid(nchunks, B * h), block(256);
    if (dtype == DEVIAS_BF16) {
        SLOTF_DISPATCH(slotf_bwd_kernel, bf16, (const bf16*)ctx, attn, rsum, (const bf16*)z, (const bf16*)dz, d_attn_ext, ds, ws, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slotf_bwd");
        hipLaunchKernelGGL((slotf_bwd_finish_kernel<bf16>), dim3(B * h * S), dim3(D <= 1024 ? (D + 63) / 64 * 64 : 256), 0, st, ws, (bf16*)dqp, S, h, D, nchunks);
    } else {
        SLOTF_DISPATCH(slotf_bwd_kernel, float, (const float*)ctx, attn, rsum, (const float*)z, (const float*)dz, d_attn_ext, ds, ws, S, N, h, scale);
        DEVIAS_CHECK_LAUNCH("devias_slotf_bwd");
        hipLaunchKernelGGL((slotf_bwd_finish_kernel<float>), dim3(B * h * S), dim3(256), 0, st, ws, (float*)dqp, S, h, D, nchunks);
    }
    DEVIAS_CHECK_LAUNCH("devias_slotf_bwd(finish)");
    return DEVIAS_OK;
}

extern "C" int devias_slotf_pack(const float* attn_stack, const float* rsum_stack, const float* ds_stack, const void* dz_stack,
                                 const void* qp_stack, void* coef, void* vec, int32_t L, int32_t B, int32_t S, int32_t N, int32_t Np,
                                 int32_t h, int32_t D, float scale, int32_t dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(attn_stack && rsum_stack && ds_stack && dz_stack && qp_stack && coef && vec, "devias_slotf_pack: null pointer");
    DEVIAS_REQUIRE(L >= 1 && B > 0 && B <= 65535 && S >= 1 && N > 0 && Np >= N && h > 0 && D > 0, "devias_slotf_pack: bad dims");
    DEVIAS_REQUIRE(dtype == DEVIAS_BF16 || dtype == DEVIAS_F32, "devias_slotf_pack: bad dtype %d", dtype);
    dim3 grid(2 * L * h * S, B), block(256);
    if (dtype == DEVIAS_BF16)
        hipLaunchKernelGGL((slotf_pack_kernel<bf16>), grid, block, 0, st, attn_stack, rsum_stack, ds_stack, (const bf16*)dz_stack,
                           (const bf16*)qp_stack, (bf16*)coef, (bf16*)vec, L, B, S, N, Np, h, D, scale);
    else
        hipLaunchKernelGGL((slotf_pack_kernel<float>), grid, block, 0, st, attn_stack, rsum_stack, ds_stack, (const float*)dz_stack,
                           (const float*)qp_stack, (float*)coef, (float*)vec, L, B, S, N, Np, h, D, scale);
    DEVIAS_CHECK_LAUNCH("devias_slotf_pack");
    return DEVIAS_OK;
}
