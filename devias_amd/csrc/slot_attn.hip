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
