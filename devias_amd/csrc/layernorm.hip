// LayerNorm forward / backward for gfx950.  One wave (64 lanes) per row, 4 elements per lane per step, the
// whole row held in registers (D <= 4096, D % 4 == 0), fp32 statistics with a two-pass (mean, then centred
// variance) reduction done by wave shuffles.  HBM-bound: x is read once, y written once.
// Backward fuses (a) dx, (b) the optional residual-branch gradient add and (c) the per-workgroup partial
// column sums for dgamma/dbeta; a second tiny kernel reduces those partials in a fixed order.
#include "common.h"
#include <string.h>

namespace {

// backward: ONE workgroup of 16 waves per CU, the rows divided evenly over the workgroups (M = 50176, D = 768, with the residual add and the dx column sums,
// tools/exp/ln_ab.py: 392 workgroups of 4 waves x 128 rows 85.5 us; 8 or 16 waves at 128 rows 85.6 / 82.2; 512 workgroups x 98 rows 77-78; 256 workgroups of 16
// waves x 196 rows 63.5 us = 4.9 TB/s -- it is the even spread over the CUs that counts, and fewer partial rows for the parameter reduce)
#ifndef DEVIAS_LNF_WGS_PER_CU
#define DEVIAS_LNF_WGS_PER_CU 8
#endif
// the backward kernel at D <= 768, bf16 (the measured step's 27 launches): waves per workgroup, rows a wave keeps in flight ahead of its current one, workgroups per CU
#ifndef DEVIAS_LNB_NW
#define DEVIAS_LNB_NW 16
#endif
#ifndef DEVIAS_LNB_PF
#define DEVIAS_LNB_PF 1
#endif
#ifndef DEVIAS_LNB_WGS
#define DEVIAS_LNB_WGS 1
#endif
enum { LN_WAVES = 4, LN_MAXIT = 16, LNB_MIN_ROWS = 32, LNB_ONE_WG_ROWS = 256, LNF_WGS_PER_CU = DEVIAS_LNF_WGS_PER_CU, LNB_NW = DEVIAS_LNB_NW, LNB_PF = DEVIAS_LNB_PF,
       LNB_WGS_PER_CU = DEVIAS_LNB_WGS };
int ln_ncu() { return devias_device_cus(); }


template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<bf16> { typedef bf16x4 type; };
__device__ __forceinline__ f32x4 raw_to_f32(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 raw_to_f32(bf16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// forward: every wave walks rows (row = first + k * stride) with the next row's loads in flight while it reduces the current one; gamma / beta stay in registers
template <typename T, int NIT>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = gridDim.x * LN_WAVES;
    int row = blockIdx.x * LN_WAVES + wave;
    if (row >= M) return;
    typedef typename Raw4<T>::type raw4;
    f32x4 g[NIT], b[NIT];
    raw4 nx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int c = (it * 64 + lane) * 4;
        if (c < D) {
            g[it] = *reinterpret_cast<const f32x4*>(gamma + c);
            b[it] = *reinterpret_cast<const f32x4*>(beta + c);
            nx[it] = *reinterpret_cast<const raw4*>(x + (int64_t)row * D + c);
        }
    }
    for (; row < M; row += stride) {
        f32x4 v[NIT];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (c < D) { v[it] = raw_to_f32(nx[it]); s += v[it][0] + v[it][1] + v[it][2] + v[it][3]; }
            else v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (row + stride < M) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int c = (it * 64 + lane) * 4;
                if (c < D) nx[it] = *reinterpret_cast<const raw4*>(x + (int64_t)(row + stride) * D + c);
            }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { float d = v[it][j] - mu; q += d * d; }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        T* yr = y + (int64_t)row * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[it][j] - mu) * rs * g[it][j] + b[it][j];
                store4(yr + c, o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}


// backward: workgroup = NW waves, each wave walks every NW-th row of the workgroup's rows; lane-owned columns are fixed so the
// dgamma/dbeta partial sums live in registers and are combined across the waves through LDS at the end.
// PF = rows a wave keeps in flight AHEAD of the one it works on (a ring of PF register sets, the row loop unrolled over it so that every set has a compile-time name).
// The kernel is a latency machine: a wave waits for a row, reduces it (two wave sums), stores it; what it has in flight meanwhile is PF rows of dy / x / dres.
// Round 3's form (16 waves, PF = 1: 16 x 4.6 KB = 74 KB per CU at best, less while waves compute) held 4.3-4.5 TB/s; Little's law at the ~3 us loaded latency of this
// part asks for >= 70 KB ALWAYS in flight to reach the 6 TB/s copy rate (MI355X_MICROARCH.md, "Indexed rows").
// FULL: D == NIT * 256 as a compile-time fact (the measured step's D = 768): no per-chunk column guards -- with them the PF = 3 form is 2700 lines of branches and spills
template <typename T, int NIT, int NW, int PF = 1, bool FULL = false>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part, int M, int D, int rows,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum, float beta_acc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [NW / 2 waves][3][NIT*256]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: the row bookkeeping lives in scalar registers)
    f32x4 g[NIT], dg[NIT], db[NIT], dc[NIT];     // dc: column sums of the stored dx
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int c = (it * 64 + lane) * 4;
        g[it] = (FULL || c < D) ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        dc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Each workgroup owns a contiguous block of `rows` rows, wave w every NW-th of them.  (Round 6, measured and not adopted -- profiles/r6_layernorm_variants.txt: the rows dealt
    // INTERLEAVED over the whole grid, wave w of workgroup b taking rows b NW + w + k (workgroups x NW) so that the chip sweeps each tensor as one sequential window like the
    // forward kernel: 58.2-59.7 us against 57.7-57.8 -- it is not the access pattern either; -DDEVIAS_LNB_INTERLEAVED keeps the form.)
#ifndef DEVIAS_LNB_INTERLEAVED
    const int rinc = NW, rend = min(M, (int)blockIdx.x * rows + rows);
    int row = blockIdx.x * rows + wave;
#else
    const int rinc = NW * gridDim.x, rend = M;
    int row = blockIdx.x * NW + wave;
#endif
    // software pipeline over rows: the (packed) loads of row r + PF * NW are issued before the reductions of row r
    typedef typename Raw4<T>::type raw4;
    raw4 nd[PF][NIT], nx[PF][NIT], nr[PF][NIT];
    float nmu[PF], nrs[PF];
    auto issue = [&](int row, raw4 (&d_)[NIT], raw4 (&x_)[NIT], raw4 (&r_)[NIT], float& mu_, float& rs_) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (FULL || c < D) {
                // (read non-temporally -- x, dy, the residual gradient, or all three: +-0.01 ms per step, in-process A/B; not kept)
                d_[it] = *reinterpret_cast<const raw4*>(dy + (int64_t)row * D + c);
                x_[it] = *reinterpret_cast<const raw4*>(x + (int64_t)row * D + c);
#if defined(DEVIAS_LNB_ABL) && (DEVIAS_LNB_ABL & 2)
                r_[it] = x_[it];
#else
                if (dres) r_[it] = *reinterpret_cast<const raw4*>(dres + (int64_t)row * D + c);
#endif
            }
        }
        mu_ = mean[row]; rs_ = rstd[row];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        nmu[s] = 0.f; nrs[s] = 0.f;
        if (row + s * rinc < rend) issue(row + s * rinc, nd[s], nx[s], nr[s], nmu[s], nrs[s]);
    }
    for (; row < rend; row += rinc * PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
        const int r = row + s * rinc;
        if (r >= rend) break;
        // pass 1 straight out of the ring set (no copies of dy / x: their registers are free for the next request as soon as the pass has read them)
        const float mu = nmu[s], rs = nrs[s];
        f32x4 a[NIT], xh[NIT];
        raw4 cr[NIT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            cr[it] = nr[s][it];
            if (FULL || c < D) {
                const f32x4 d = raw_to_f32(nd[s][it]);
                const f32x4 xv = raw_to_f32(nx[s][it]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float h = (xv[j] - mu) * rs;
                    xh[it][j] = h;
                    dg[it][j] += d[j] * h;
                    db[it][j] += d[j];
                    float aj = d[j] * g[it][j];
                    a[it][j] = aj;
                    s1 += aj; s2 += aj * h;
                }
            } else { a[it] = f32x4{0.f, 0.f, 0.f, 0.f}; xh[it] = a[it]; }
        }
        if (r + PF * rinc < rend) issue(r + PF * rinc, nd[s], nx[s], nr[s], nmu[s], nrs[s]);
#if defined(DEVIAS_LNB_ABL) && (DEVIAS_LNB_ABL & 4)      // (diagnostic builds only, results wrong on purpose: tools/exp/ln_ab.py)
        const float m1 = s1 / (float)D, m2 = s2 / (float)D;
#else
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#endif
        T* dxr = dx + (int64_t)r * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (FULL || c < D) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (a[it][j] - m1 - xh[it][j] * m2);
                if (dres) o += raw_to_f32(cr[it]);
                dc[it] += o;
#if defined(DEVIAS_LNB_ABL) && (DEVIAS_LNB_ABL & 1)
                if (o[0] == 12345.678f)
#endif
                store4(dxr + c, o);
            }
        }
        }
    }
    // combine the waves' partials in a fixed tree (of the n waves left, waves [h, n) hand theirs to waves [0, n - h) through LDS, h = ceil(n / 2); n = NW ... 2) -> part[blockIdx][3][D]
    const int W = NIT * 256;
#pragma unroll
    for (int n = NW; n > 1; n = (n + 1) / 2) {
        const int h = (n + 1) / 2;
        if (wave >= h && wave < n) {
            float* dst = sm + (size_t)(wave - h) * 3 * W;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int c = (it * 64 + lane) * 4;
                *reinterpret_cast<f32x4*>(dst + c) = dg[it];
                *reinterpret_cast<f32x4*>(dst + W + c) = db[it];
                *reinterpret_cast<f32x4*>(dst + 2 * W + c) = dc[it];
            }
        }
        __syncthreads();
        if (wave < n - h) {
            const float* src = sm + (size_t)wave * 3 * W;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int c = (it * 64 + lane) * 4;
                dg[it] += *reinterpret_cast<const f32x4*>(src + c);
                db[it] += *reinterpret_cast<const f32x4*>(src + W + c);
                dc[it] += *reinterpret_cast<const f32x4*>(src + 2 * W + c);
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* out = part + (int64_t)blockIdx.x * 3 * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c = (it * 64 + lane) * 4;
            if (FULL || c < D) {
                const f32x4 a = dg[it], b = db[it], cc = dc[it];
                if (dgamma) {      // a single workgroup: these ARE the results -- no reduce pass (same values: the pass adds zeros to them)
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(dgamma + c) = a + (beta_acc != 0.f ? beta_acc * *reinterpret_cast<const f32x4*>(dgamma + c) : z);
                    *reinterpret_cast<f32x4*>(dbeta + c) = b + (beta_acc != 0.f ? beta_acc * *reinterpret_cast<const f32x4*>(dbeta + c) : z);
                    if (dxsum) *reinterpret_cast<f32x4*>(dxsum + c) = cc + z;
                    continue;
                }
                *reinterpret_cast<f32x4*>(out + c) = a;
                *reinterpret_cast<f32x4*>(out + D + c) = b;
                *reinterpret_cast<f32x4*>(out + 2 * D + c) = cc;
            }
        }
    }
}

// out[i] (i over 3*D: dgamma | dbeta | dx column sums) = beta_acc*out[i] + sum_p part[p][i]; block (64 columns, 16 partial lanes)
__global__ void ln_param_reduce_kernel(const float* __restrict__ part, int nparts, int D, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ dxsum, float beta_acc) {
    __shared__ float sm[16][64];
    const int lim = dxsum ? 3 * D : 2 * D;
    int i = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (i < lim)
        for (int p = threadIdx.y; p < nparts; p += 16) s += part[(int64_t)p * 3 * D + i];
    sm[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && i < lim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][threadIdx.x];
        if (i >= 2 * D) { dxsum[i - 2 * D] = t; return; }
        float* dst = i < D ? dgamma + i : dbeta + (i - D);
        *dst = t + (beta_acc != 0.f ? beta_acc * *dst : 0.f);
    }
}

template <typename T>
int ln_fwd_dispatch(const T* x, const float* g, const float* b, T* y, float* mean, float* rstd, int M, int D, float eps,
                    hipStream_t st) {
    // rows per wave: at most DEVIAS_LNF_WGS_PER_CU workgroups of 4 waves per CU walk the rows (tools/exp/ln_ab.py)
    const int cap = ln_ncu() * LNF_WGS_PER_CU;
    dim3 grid(cdiv(M, LN_WAVES) < cap ? cdiv(M, LN_WAVES) : cap), block(256);
    int nit = cdiv(D, 256);
#define LNF(N) hipLaunchKernelGGL((ln_fwd_kernel<T, N>), grid, block, 0, st, x, g, b, y, mean, rstd, M, D, eps)
    if (nit <= 2) LNF(2); else if (nit <= 3) LNF(3); else if (nit <= 4) LNF(4); else if (nit <= 8) LNF(8); else LNF(16);
#undef LNF
    return 0;
}
// workgroups and rows per workgroup of the backward kernel: up to LNB_ONE_WG_ROWS rows one workgroup (which then writes the parameter gradients itself),
// otherwise the rows spread evenly over at most one workgroup per CU
// (the CU count is the policy's: device CUs minus the option gemm_reserve_cus -- a workgroup of this kernel fills a CU, so with K CUs held by a concurrent kernel (RCCL during
// backward) K workgroups of a one-per-CU grid would run as a second round: twice the launch's time, +1.3 ms per step with 16 held, profiles/r6_cu_hog.txt)
void ln_bwd_shape(int M, int& nwg, int& rows) {
    const int cus = devias_policy_gemm_cus() * LNB_WGS_PER_CU;
    nwg = M <= LNB_ONE_WG_ROWS ? 1 : (cdiv(M, LNB_MIN_ROWS) < cus ? cdiv(M, LNB_MIN_ROWS) : cus);
    rows = cdiv(M, nwg);
    nwg = cdiv(M, rows);
}

template <typename T>
int ln_bwd_dispatch(const T* dy, const T* x, const float* g, const float* mean, const float* rstd, const T* dres, T* dx,
                    float* part, int M, int D, float* dgamma, float* dbeta, float* dxsum, float beta_acc, hipStream_t st) {
    int nwg, rows;
    ln_bwd_shape(M, nwg, rows);
    int nit = cdiv(D, 256);
    // waves per workgroup: 16 where the row fits 128 registers per lane (bf16 up to D = 768, fp32 up to 512), 8 up to D = 1024, 4 beyond: no instantiation may spill
    // (the combine tree's LDS is NW / 2 x 3 x NIT x 1 KiB)
    constexpr bool half = sizeof(T) == 2;
#define LNB(N, NW, PF, FULL) hipLaunchKernelGGL((ln_bwd_kernel<T, N, NW, PF, FULL>), dim3(nwg), dim3(NW * 64), ((NW + 1) / 2) * 3 * N * 256 * sizeof(float), st, dy, x, g, mean, rstd, dres, dx, part, M, D, rows, \
                                      dgamma, dbeta, dxsum, beta_acc)
    if (nit <= 2) LNB(2, 16, 1, false);
    else if (nit <= 3) { if constexpr (half) { if (D == 768) LNB(3, LNB_NW, LNB_PF, (LNB_NW != 16)); else LNB(3, 16, 1, false); } else LNB(3, 8, 1, false); }
    else if (nit <= 4) LNB(4, 8, 1, false); else LNB(8, 4, 1, false);
#undef LNB
    return 0;
}

}  // namespace

// LayerNorm backward of a FEW rows (the B * S slot rows of the aggregation block), partials only: workgroups of four waves, one row per wave, write
// part[workgroup][3][D] (dgamma | dbeta | dx column sums) and nobody reduces them -- the caller stacks the partials of every layer that shares the parameters and
// runs ONE fixed-order reduce per parameter at the end (csrc/regions.hip, devias_agg_block_bwd).  The single-workgroup form of devias_layernorm_bwd, which writes
// the results itself, is 16 waves sharing one CU's four SIMDs for four rows each: 13.5 us per call, 18 calls per step.
enum { LNB_SMALL_ROWS = 4 };
int64_t devias_layernorm_bwd_parts_count(int M) { return cdiv(M, LNB_SMALL_ROWS); }
int devias_layernorm_bwd_parts(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, float* part,
                               int M, int D, int dtype, hipStream_t st) {
    DEVIAS_REQUIRE(dy && x && gamma && mean && rstd && dx && part && M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "devias_layernorm_bwd_parts: bad args");
    const int nwg = cdiv(M, LNB_SMALL_ROWS), nit = cdiv(D, 256);
#define LNBS(T, N) hipLaunchKernelGGL((ln_bwd_kernel<T, N, 4>), dim3(nwg), dim3(4 * 64), (4 / 2) * 3 * N * 256 * sizeof(float), st, (const T*)dy, (const T*)x, gamma, mean, rstd, \
                                      (const T*)dres, (T*)dx, part, M, D, (int)LNB_SMALL_ROWS, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f)
    if (dtype == DEVIAS_BF16) { if (nit <= 2) LNBS(bf16, 2); else if (nit <= 3) LNBS(bf16, 3); else if (nit <= 4) LNBS(bf16, 4); else LNBS(bf16, 8); }
    else if (dtype == DEVIAS_F32) { if (nit <= 2) LNBS(float, 2); else if (nit <= 3) LNBS(float, 3); else if (nit <= 4) LNBS(float, 4); else LNBS(float, 8); }
    else return devias_set_error(DEVIAS_EINVAL, "devias_layernorm_bwd_parts: bad dtype %d", dtype);
#undef LNBS
    DEVIAS_CHECK_LAUNCH("devias_layernorm_bwd_parts");
    return DEVIAS_OK;
}

extern "C" int devias_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, int32_t M, int32_t D, float eps, int32_t dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(x && gamma && beta && y && mean && rstd, "devias_layernorm_fwd: null pointer");
    DEVIAS_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, "devias_layernorm_fwd: need D %% 4 == 0 and D <= 4096 (D=%d)", D);
    DEVIAS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), "devias_layernorm_fwd: unaligned pointer");
    if (dtype == DEVIAS_BF16) ln_fwd_dispatch<bf16>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, eps, st);
    else if (dtype == DEVIAS_F32) ln_fwd_dispatch<float>((const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps, st);
    else return devias_set_error(DEVIAS_EINVAL, "devias_layernorm_fwd: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_layernorm_fwd");
    return DEVIAS_OK;
}

extern "C" int64_t devias_layernorm_bwd_workspace_bytes(int32_t M, int32_t D) {
    int nwg, rows;
    ln_bwd_shape(M, nwg, rows);
    return (int64_t)nwg * 3 * D * 4;
}

extern "C" int devias_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                    const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                    float beta_acc, float* dx_colsum, int32_t M, int32_t D, int32_t dtype, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, "devias_layernorm_bwd: null pointer");
    DEVIAS_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "devias_layernorm_bwd: need D %% 4 == 0 and D <= 2048 (D=%d)", D);
    DEVIAS_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(gamma) && (!dres || aligned16(dres)) && aligned16(ws),
                   "devias_layernorm_bwd: unaligned pointer");
    int nparts, rows_per_wg;
    ln_bwd_shape(M, nparts, rows_per_wg);
    const bool direct = nparts == 1 && aligned16(dgamma) && aligned16(dbeta) && (!dx_colsum || aligned16(dx_colsum));   // one workgroup: it writes the results itself
    float *dg_k = direct ? dgamma : nullptr, *db_k = direct ? dbeta : nullptr, *ds_k = direct ? dx_colsum : nullptr;
    if (dtype == DEVIAS_BF16)
        ln_bwd_dispatch<bf16>((const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, ws, M, D, dg_k, db_k, ds_k, beta_acc, st);
    else if (dtype == DEVIAS_F32)
        ln_bwd_dispatch<float>((const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)dres, (float*)dx, ws, M, D, dg_k, db_k, ds_k, beta_acc, st);
    else return devias_set_error(DEVIAS_EINVAL, "devias_layernorm_bwd: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_layernorm_bwd");
    if (direct) return DEVIAS_OK;
    {   // dgamma | dbeta | (dx column sums): three second stages over the [nparts][3D] partials
        const DeviasReduceJob j[3] = {{ws, nparts, 3 * D, D, dgamma, beta_acc}, {ws + D, nparts, 3 * D, D, dbeta, beta_acc}, {ws + 2 * D, nparts, 3 * D, D, dx_colsum, 0.f}};
        if (devias_defer(j, dx_colsum ? 3 : 2)) return DEVIAS_OK;
    }
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(3 * D, 64)), dim3(64, 16), 0, st, ws, nparts, D, dgamma, dbeta, dx_colsum, beta_acc);
    DEVIAS_CHECK_LAUNCH("devias_layernorm_bwd(param reduce)");
    return DEVIAS_OK;
}
