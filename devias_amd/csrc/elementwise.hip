// HBM-bound helpers: casts, tubelet im2col, column sums (bias gradients), row broadcast / reduce, add, AdamW.
// All are vectorised to 16 bytes per lane where alignment allows and grid-stride over <= 2048 workgroups.
#include "common.h"

namespace {

__device__ __forceinline__ float ld_as_f32(const void* p, int dt, int64_t i) {
    return dt == DEVIAS_BF16 ? (float)reinterpret_cast<const bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_from_f32(void* p, int dt, int64_t i, float v) {
    if (dt == DEVIAS_BF16) reinterpret_cast<bf16*>(p)[i] = (bf16)v;
    else reinterpret_cast<float*>(p)[i] = v;
}

// ---- cast -------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n, int vec) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        int64_t n4 = n >> 2;
        for (int64_t i = i0; i < n4; i += stride) store4(dst + 4 * i, load4(src + 4 * i));
        for (int64_t i = (n4 << 2) + i0; i < n; i += stride) dst[i] = from_f32<D>(to_f32(src[i]));
    } else {
        for (int64_t i = i0; i < n; i += stride) dst[i] = from_f32<D>(to_f32(src[i]));
    }
}

// (no __restrict__: dst may be src -- devias_cast_scale's contract, used in place by GradSync.finish(); every thread reads and writes the same index)
template <typename S, typename D>
__global__ void cast_scale_kernel(const S* src, D* dst, int64_t n, int vec, float scale) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        int64_t n4 = n >> 2;
        for (int64_t i = i0; i < n4; i += stride) store4(dst + 4 * i, load4(src + 4 * i) * scale);
        for (int64_t i = (n4 << 2) + i0; i < n; i += stride) dst[i] = from_f32<D>(to_f32(src[i]) * scale);
    } else {
        for (int64_t i = i0; i < n; i += stride) dst[i] = from_f32<D>(to_f32(src[i]) * scale);
    }
}

// ---- im2col -----------------------------------------------------------------------------------------
// one thread moves 4 consecutive kw pixels (ps % 4 == 0): 16-byte fp32 reads, 8-byte bf16 writes
template <typename S, typename D>
__global__ void im2col_kernel(const S* __restrict__ x, D* __restrict__ out, int B, int C, int T, int H, int W,
                              int ts, int ps) {
    const int g_h = H / ps, g_w = W / ps, Tp = T / ts;
    const int F = C * ts * ps * ps;              // features per token
    const int64_t total4 = (int64_t)B * Tp * g_h * g_w * F / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = i * 4;
        int f = (int)(e % F);
        int64_t tok = e / F;
        int kw = f % ps; int r = f / ps;
        int kh = r % ps; r /= ps;
        int kt = r % ts; int c = r / ts;
        int w_ = (int)(tok % g_w); int64_t r2 = tok / g_w;
        int h_ = (int)(r2 % g_h); r2 /= g_h;
        int t_ = (int)(r2 % Tp); int b = (int)(r2 / Tp);
        int64_t src = ((((int64_t)b * C + c) * T + (t_ * ts + kt)) * H + (h_ * ps + kh)) * W + (w_ * ps + kw);
        store4(out + e, load4(x + src));
    }
}

// ---- column sums --------------------------------------------------------------------------------------
// stage 1: workgroup = 32 column lanes x 8 row lanes; each lane owns 8 consecutive columns (one 16-byte bf16 / two 16-byte
// fp32 loads per row), so a wave reads 2 rows x 512 contiguous bytes per instruction.  CS_ROWS rows per workgroup ->
// partial[rb][n]; stage 2: out[n] = beta*out[n] + sum_rb partial (fixed order).
enum { CS_ROWS = 256, CS_COLS = 256 };
// SCALE: y = x * scale[row / rps] is written as well (devias_row_scale's arithmetic) and the sums are those of the STORED (rounded) y: one pass instead of the
// row-scale pass followed by a column-sum pass over its output, bitwise the same results (vector path only: the host falls back to the two kernels otherwise)
__device__ __forceinline__ f32x4 stored4(f32x4 v, const float*) { return v; }
__device__ __forceinline__ f32x4 stored4(f32x4 v, const bf16*) { return f32x4{(float)(bf16)v[0], (float)(bf16)v[1], (float)(bf16)v[2], (float)(bf16)v[3]}; }
template <typename T, bool SCALE = false>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int M, int N, int ldx, float* __restrict__ part, int vec,
                                                             float* __restrict__ out, float beta, const float* __restrict__ scale = nullptr, int rps = 1,
                                                             T* __restrict__ y = nullptr) {
    __shared__ float sm[8][CS_COLS + 8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n0 = blockIdx.x * CS_COLS + tx * 8;
    const int r0 = blockIdx.y * CS_ROWS;
    const int r1 = min(M, r0 + CS_ROWS);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (vec && n0 + 7 < N) {
        for (int r = r0 + ty; r < r1; r += 8) {
            f32x4 a = load4(x + (int64_t)r * ldx + n0), b = load4(x + (int64_t)r * ldx + n0 + 4);
            if constexpr (SCALE) {
                const float sc = scale[r / rps];
                a = a * sc; b = b * sc;
                store4(y + (int64_t)r * ldx + n0, a); store4(y + (int64_t)r * ldx + n0 + 4, b);
                a = stored4(a, x); b = stored4(b, x);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += a[j]; s[4 + j] += b[j]; }
        }
    } else {
        for (int r = r0 + ty; r < r1; r += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) if (n0 + j < N) s[j] += to_f32(x[(int64_t)r * ldx + n0 + j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[ty][tx * 8 + j] = s[j];
    __syncthreads();
    const int c = threadIdx.x;              // 256 threads <-> 256 columns
    const int n = blockIdx.x * CS_COLS + c;
    if (n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][c];
        if (out) out[n] = t + (beta != 0.f ? beta * out[n] : 0.f);      // a single row block (M <= CS_ROWS): the result itself, no second stage
        else part[(int64_t)blockIdx.y * N + n] = t;
    }
}
// block (64 columns, 16 partial lanes)
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int N, float* __restrict__ out, float beta) {
    __shared__ float sm[16][64];
    int n = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (n < N)
        for (int i = threadIdx.y; i < nparts; i += 16) s += part[(int64_t)i * N + n];
    sm[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][threadIdx.x];
        out[n] = t + (beta != 0.f ? beta * out[n] : 0.f);
    }
}

template <typename T>
__global__ void rows_reduce_mod_kernel(const T* __restrict__ x, int M, int N, int mod, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mod * N) return;
    int r0 = i / N, n = i % N;
    float s = 0.f;
    for (int r = r0; r < M; r += mod) s += to_f32(x[(int64_t)r * N + n]);
    out[i] = s;
}
template <typename T>
__global__ void rows_broadcast_kernel(const float* __restrict__ src, int mod, int N, T* __restrict__ out, int M) {
    int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / N), n = (int)(i % N);
        out[i] = from_f32<T>(src[(int64_t)(r % mod) * N + n]);
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n, int vec) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        int64_t n4 = n >> 2;
        for (int64_t i = i0; i < n4; i += stride) store4(y + 4 * i, load4(a + 4 * i) + load4(b + 4 * i));
        for (int64_t i = (n4 << 2) + i0; i < n; i += stride) y[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
    } else {
        for (int64_t i = i0; i < n; i += stride) y[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
    }
}

// y = a * mask (+ b): element-wise dropout (mask = 0 or 1/keep, fp32) and its backward with the gradient fan-in of the un-dropped consumers
template <typename T>
__global__ void mul_mask_kernel(const T* __restrict__ a, const float* __restrict__ mask, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = to_f32(a[i]) * mask[i];
        if (b) v += to_f32(b[i]);
        y[i] = from_f32<T>(v);
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = to_f32(dy[i]), v = to_f32(y[i]);
        float r = act == DEVIAS_ACT_SIGMOID ? g * v * (1.0f - v) : act == DEVIAS_ACT_RELU ? (v > 0.f ? g : 0.f) : g * dgelu_t<T>(v);
        dx[i] = from_f32<T>(r);
    }
}

template <typename T>
__global__ void row_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale, int rps, T* __restrict__ y, int M, int N) {
    const int64_t total4 = (int64_t)M * N / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i * 4 / N);
        store4(y + i * 4, load4(x + i * 4) * scale[m / rps]);
    }
}

// ---- AdamW ------------------------------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        float pi = p[i] * (1.0f - lr * wd);                      // decoupled weight decay (torch.optim.AdamW)
        float mi = m[i] + (1.0f - b1) * (gi - m[i]);             // exp_avg.lerp_(grad, 1-beta1)
        float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}

// ---- multi-tensor optimizer: one workgroup per DEVIAS_OPT_CHUNK-element piece of some tensor -------------------------
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const devias_opt_tensor* __restrict__ table, const int32_t* __restrict__ ct,
                                                          const int32_t* __restrict__ ci, float* __restrict__ partials) {
    __shared__ float red[4];
    const devias_opt_tensor t = table[ct[blockIdx.x]];
    const int64_t lo = (int64_t)ci[blockIdx.x] * DEVIAS_OPT_CHUNK;
    const int64_t hi = lo + DEVIAS_OPT_CHUNK < t.n ? lo + DEVIAS_OPT_CHUNK : t.n;
    const float* g = t.grad + lo;
    const int len = (int)(hi - lo);
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int n4 = len >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 v = reinterpret_cast<const float4*>(g)[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < len; i += 256) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < len; i += 256) s += g[i] * g[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;                                               // fixed order: thread t owns partials t, t+256, ...
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float norm = sqrtf(red[0]);
        out[0] = norm;
        float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        out[1] = c < 1.f ? c : 1.f;
    }
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const devias_opt_tensor* __restrict__ table, const int32_t* __restrict__ ct,
                                                          const int32_t* __restrict__ ci, float b1, float b2, float eps, float gscale,
                                                          const float* __restrict__ gscale_dev) {
    const devias_opt_tensor t = table[ct[blockIdx.x]];
    const int64_t lo = (int64_t)ci[blockIdx.x] * DEVIAS_OPT_CHUNK;
    const int64_t hi = lo + DEVIAS_OPT_CHUNK < t.n ? lo + DEVIAS_OPT_CHUNK : t.n;
    const int len = (int)(hi - lo);
    if (gscale_dev) gscale *= *gscale_dev;
    float* p = t.param + lo; const float* g = t.grad + lo; float* m = t.exp_avg + lo; float* v = t.exp_avg_sq + lo;
    const float decay = 1.0f - t.lr * t.weight_decay, step = t.lr / t.bc1, ib2 = 1.0f / t.bc2_sqrt;
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= gscale;
        pi *= decay;                                              // decoupled weight decay (torch.optim.AdamW)
        mi = mi + (1.0f - b1) * (gi - mi);                        // exp_avg.lerp_(grad, 1-beta1)
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        pi = pi - step * (mi / (sqrtf(vi) * ib2 + eps));
    };
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    int done = 0;
    if (al) {
        const int n4 = len >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
            float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
            reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < len; i += 256) upd(p[i], g[i], m[i], v[i]);
}

inline int grid_for(int64_t n, int per_thread = 1) {
    int64_t b = (n / per_thread + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace

extern "C" int devias_cast(const void* src, int32_t sd, void* dst, int32_t dd, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(src && dst && n >= 0, "devias_cast: bad args");
    if (n == 0) return DEVIAS_OK;
    int vec = aligned16(src) && aligned16(dst);
    dim3 g(grid_for(n, 4)), b(256);
    if (sd == DEVIAS_F32 && dd == DEVIAS_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, st, (const float*)src, (bf16*)dst, n, vec);
    else if (sd == DEVIAS_BF16 && dd == DEVIAS_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, st, (const bf16*)src, (float*)dst, n, vec);
    else if (sd == DEVIAS_F32 && dd == DEVIAS_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n, vec);
    else if (sd == DEVIAS_BF16 && dd == DEVIAS_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, st, (const bf16*)src, (bf16*)dst, n, vec);
    else return devias_set_error(DEVIAS_EINVAL, "devias_cast: bad dtypes %d -> %d", sd, dd);
    DEVIAS_CHECK_LAUNCH("devias_cast");
    return DEVIAS_OK;
}

extern "C" int devias_cast_scale(const void* src, int32_t sd, void* dst, int32_t dd, int64_t n, float scale, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(src && dst && n >= 0, "devias_cast_scale: bad args");
    if (n == 0) return DEVIAS_OK;
    int vec = aligned16(src) && aligned16(dst);
    dim3 g(grid_for(n, 4)), b(256);
    if (sd == DEVIAS_F32 && dd == DEVIAS_F32) hipLaunchKernelGGL((cast_scale_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n, vec, scale);
    else if (sd == DEVIAS_BF16 && dd == DEVIAS_F32) hipLaunchKernelGGL((cast_scale_kernel<bf16, float>), g, b, 0, st, (const bf16*)src, (float*)dst, n, vec, scale);
    else if (sd == DEVIAS_F32 && dd == DEVIAS_BF16) hipLaunchKernelGGL((cast_scale_kernel<float, bf16>), g, b, 0, st, (const float*)src, (bf16*)dst, n, vec, scale);
    else return devias_set_error(DEVIAS_EINVAL, "devias_cast_scale: bad dtypes %d -> %d", sd, dd);
    DEVIAS_CHECK_LAUNCH("devias_cast_scale");
    return DEVIAS_OK;
}

extern "C" int devias_patch_im2col(const void* x, int32_t xd, void* out, int32_t od, int32_t B, int32_t C, int32_t T,
                                   int32_t H, int32_t W, int32_t ts, int32_t ps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(x && out, "devias_patch_im2col: null pointer");
    DEVIAS_REQUIRE(ps % 4 == 0 && H % ps == 0 && W % ps == 0 && T % ts == 0 && W % 4 == 0,
                   "devias_patch_im2col: need ps%%4==0 and H,W divisible by ps, T by ts (got H=%d W=%d T=%d ps=%d ts=%d)", H, W, T, ps, ts);
    DEVIAS_REQUIRE(aligned16(x) && aligned16(out), "devias_patch_im2col: pointers must be 16-byte aligned");
    int64_t total4 = (int64_t)B * C * T * H * W / 4;
    dim3 g(grid_for(total4)), b(256);
    if (xd == DEVIAS_F32 && od == DEVIAS_F32) hipLaunchKernelGGL((im2col_kernel<float, float>), g, b, 0, st, (const float*)x, (float*)out, B, C, T, H, W, ts, ps);
    else if (xd == DEVIAS_F32 && od == DEVIAS_BF16) hipLaunchKernelGGL((im2col_kernel<float, bf16>), g, b, 0, st, (const float*)x, (bf16*)out, B, C, T, H, W, ts, ps);
    else if (xd == DEVIAS_BF16 && od == DEVIAS_BF16) hipLaunchKernelGGL((im2col_kernel<bf16, bf16>), g, b, 0, st, (const bf16*)x, (bf16*)out, B, C, T, H, W, ts, ps);
    else if (xd == DEVIAS_BF16 && od == DEVIAS_F32) hipLaunchKernelGGL((im2col_kernel<bf16, float>), g, b, 0, st, (const bf16*)x, (float*)out, B, C, T, H, W, ts, ps);
    else return devias_set_error(DEVIAS_EINVAL, "devias_patch_im2col: bad dtypes");
    DEVIAS_CHECK_LAUNCH("devias_patch_im2col");
    return DEVIAS_OK;
}

// every deferred second stage of a region in one launch: blockIdx.y = job; the loop and the combine are those of colsum_final_kernel /
// gemm_colsum_final_kernel / ln_param_reduce_kernel, so the results are bitwise theirs
struct ReduceJobs { DeviasReduceJob j[DeviasDeferList::MAX]; };
__global__ void reduce_jobs_kernel(ReduceJobs jobs) {
    __shared__ float sm[16][64];
    const DeviasReduceJob& jb = jobs.j[blockIdx.y];
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= jb.n) return;                   // (whole block past this job's width)
    float s = 0.f;
    if (n < jb.n) {
        // eight loads in flight, added in index order (the same sums as the one-at-a-time loop: this is a latency chain of nparts / 16 dependent loads otherwise --
        // 26 of them for the 416 partials of the attention backward's bias gradients)
        int i = threadIdx.y;
        for (; i + 7 * 16 < jb.nparts; i += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = jb.part[(int64_t)(i + 16 * u) * jb.stride + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < jb.nparts; i += 16) s += jb.part[(int64_t)i * jb.stride + n];
    }
    sm[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && n < jb.n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][threadIdx.x];
        jb.out[n] = t + (jb.beta != 0.f ? jb.beta * jb.out[n] : 0.f);
    }
}
int devias_flush_deferred(DeviasDeferList* l, hipStream_t st) {
    if (!l || l->n == 0) return DEVIAS_OK;
    ReduceJobs jobs;
    int nmax = 0;
    for (int i = 0; i < l->n; ++i) { jobs.j[i] = l->jobs[i]; if (l->jobs[i].n > nmax) nmax = l->jobs[i].n; }
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(cdiv(nmax, 64), l->n), dim3(64, 16), 0, st, jobs);
    l->n = 0;
    DEVIAS_CHECK_LAUNCH("devias_flush_deferred");
    return DEVIAS_OK;
}

extern "C" int64_t devias_colsum_workspace_bytes(int32_t M, int32_t N) { return (int64_t)cdiv(M, CS_ROWS) * N * 4; }

// second stage of a column sum whose [nparts, N] fp32 partials some kernel has written: out = beta * out + sum over the parts in index order.  Taken over by the
// collecting region when there is one (common.h: deferred final reductions), else its own launch.
int devias_colsum_finish(const float* part, int nparts, int N, float* out, float beta, hipStream_t st) {
    { const DeviasReduceJob j = {part, nparts, N, N, out, beta}; if (devias_defer(&j, 1)) return DEVIAS_OK; }
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 64)), dim3(64, 16), 0, st, part, nparts, N, out, beta);
    DEVIAS_CHECK_LAUNCH("devias_colsum(final)");
    return DEVIAS_OK;
}

extern "C" int devias_colsum(const void* x, int32_t dtype, int32_t M, int32_t N, int32_t ldx, float* out, float beta,
                             float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(x && out && ws && M > 0 && N > 0, "devias_colsum: bad args");
    int nparts = cdiv(M, CS_ROWS);
    dim3 g(cdiv(N, CS_COLS), nparts), b(256);
    int vec = (ldx % 8 == 0) && aligned16(x);
    float* direct = nparts == 1 ? out : nullptr;
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((colsum_partial_kernel<bf16>), g, b, 0, st, (const bf16*)x, M, N, ldx, ws, vec, direct, beta);
    else hipLaunchKernelGGL((colsum_partial_kernel<float>), g, b, 0, st, (const float*)x, M, N, ldx, ws, vec, direct, beta);
    DEVIAS_CHECK_LAUNCH("devias_colsum(partial)");
    if (direct) return DEVIAS_OK;
    return devias_colsum_finish(ws, nparts, N, out, beta, st);
}

extern "C" int devias_rows_reduce_mod(const void* x, int32_t dtype, int32_t M, int32_t N, int32_t mod, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(x && out && mod > 0 && M % mod == 0, "devias_rows_reduce_mod: bad args (M=%d mod=%d)", M, mod);
    dim3 g(cdiv((int64_t)mod * N, 256)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((rows_reduce_mod_kernel<bf16>), g, b, 0, st, (const bf16*)x, M, N, mod, out);
    else hipLaunchKernelGGL((rows_reduce_mod_kernel<float>), g, b, 0, st, (const float*)x, M, N, mod, out);
    DEVIAS_CHECK_LAUNCH("devias_rows_reduce_mod");
    return DEVIAS_OK;
}

extern "C" int devias_rows_broadcast(const float* src, int32_t mod, int32_t N, void* out, int32_t dtype, int32_t M, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(src && out && mod > 0, "devias_rows_broadcast: bad args");
    dim3 g(grid_for((int64_t)M * N)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((rows_broadcast_kernel<bf16>), g, b, 0, st, src, mod, N, (bf16*)out, M);
    else hipLaunchKernelGGL((rows_broadcast_kernel<float>), g, b, 0, st, src, mod, N, (float*)out, M);
    DEVIAS_CHECK_LAUNCH("devias_rows_broadcast");
    return DEVIAS_OK;
}

extern "C" int devias_row_scale(const void* x, const float* scale, int32_t rps, void* y, int32_t dtype, int32_t M, int32_t N, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(x && scale && y && rps > 0 && M > 0 && N > 0 && N % 4 == 0, "devias_row_scale: bad args (N must be a multiple of 4)");
    DEVIAS_REQUIRE(aligned16(x) && aligned16(y), "devias_row_scale: unaligned pointer");
    dim3 g(grid_for((int64_t)M * N / 4)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((row_scale_kernel<bf16>), g, b, 0, st, (const bf16*)x, scale, rps, (bf16*)y, M, N);
    else if (dtype == DEVIAS_F32) hipLaunchKernelGGL((row_scale_kernel<float>), g, b, 0, st, (const float*)x, scale, rps, (float*)y, M, N);
    else return devias_set_error(DEVIAS_EINVAL, "devias_row_scale: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_row_scale");
    return DEVIAS_OK;
}

// y = x * scale[row / rps] AND out = beta * out + column sums of y, in one pass over x (the stochastic-depth backward of an encoder block: the rescaled branch
// gradient and its bias gradient).  Bitwise devias_row_scale followed by devias_colsum of its output.  ws: devias_colsum_workspace_bytes(M, N).
int devias_row_scale_colsum(const void* x, const float* scale, int rps, void* y, int dtype, int M, int N, float* out, float beta, float* ws, hipStream_t st) {
    DEVIAS_REQUIRE(x && scale && y && out && ws && rps > 0 && M > 0 && N > 0, "devias_row_scale_colsum: bad args");
    if (N % 8 != 0 || !aligned16(x) || !aligned16(y) || M <= CS_ROWS) {                    // (no vector path / single row block: the two kernels)
        int rc = devias_row_scale(x, scale, rps, y, dtype, M, N, st);
        return rc ? rc : devias_colsum(y, dtype, M, N, N, out, beta, ws, st);
    }
    const int nparts = cdiv(M, CS_ROWS);
    dim3 g(cdiv(N, CS_COLS), nparts), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((colsum_partial_kernel<bf16, true>), g, b, 0, st, (const bf16*)x, M, N, N, ws, 1, (float*)nullptr, beta, scale, rps, (bf16*)y);
    else if (dtype == DEVIAS_F32) hipLaunchKernelGGL((colsum_partial_kernel<float, true>), g, b, 0, st, (const float*)x, M, N, N, ws, 1, (float*)nullptr, beta, scale, rps, (float*)y);
    else return devias_set_error(DEVIAS_EINVAL, "devias_row_scale_colsum: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_row_scale_colsum");
    return devias_colsum_finish(ws, nparts, N, out, beta, st);
}

extern "C" int devias_add(const void* a, const void* b_, void* y, int32_t dtype, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(a && b_ && y, "devias_add: null pointer");
    if (n == 0) return DEVIAS_OK;
    int vec = aligned16(a) && aligned16(b_) && aligned16(y);
    dim3 g(grid_for(n, 4)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((add_kernel<bf16>), g, b, 0, st, (const bf16*)a, (const bf16*)b_, (bf16*)y, n, vec);
    else hipLaunchKernelGGL((add_kernel<float>), g, b, 0, st, (const float*)a, (const float*)b_, (float*)y, n, vec);
    DEVIAS_CHECK_LAUNCH("devias_add");
    return DEVIAS_OK;
}

extern "C" int devias_mul_mask(const void* a, const float* mask, const void* b_, void* y, int32_t dtype, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(a && mask && y, "devias_mul_mask: null pointer");
    if (n == 0) return DEVIAS_OK;
    dim3 g(grid_for(n)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((mul_mask_kernel<bf16>), g, b, 0, st, (const bf16*)a, mask, (const bf16*)b_, (bf16*)y, n);
    else if (dtype == DEVIAS_F32) hipLaunchKernelGGL((mul_mask_kernel<float>), g, b, 0, st, (const float*)a, mask, (const float*)b_, (float*)y, n);
    else return devias_set_error(DEVIAS_EINVAL, "devias_mul_mask: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_mul_mask");
    return DEVIAS_OK;
}

extern "C" int devias_act_bwd(const void* dy, const void* y, void* dx, int32_t act, int32_t dtype, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(dy && y && dx, "devias_act_bwd: null pointer");
    DEVIAS_REQUIRE(act == DEVIAS_ACT_SIGMOID || act == DEVIAS_ACT_RELU || act == DEVIAS_ACT_GELU, "devias_act_bwd: bad act %d", act);
    if (n == 0) return DEVIAS_OK;
    dim3 g(grid_for(n)), b(256);
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16>), g, b, 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n, act);
    else if (dtype == DEVIAS_F32) hipLaunchKernelGGL((act_bwd_kernel<float>), g, b, 0, st, (const float*)dy, (const float*)y, (float*)dx, n, act);
    else return devias_set_error(DEVIAS_EINVAL, "devias_act_bwd: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_act_bwd");
    return DEVIAS_OK;
}

extern "C" int devias_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                                 void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "devias_adamw_step: bad args");
    if (n == 0) return DEVIAS_OK;
    float bc1 = 1.0f - powf(beta1, (float)step);
    float bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
    DEVIAS_CHECK_LAUNCH("devias_adamw_step");
    return DEVIAS_OK;
}

extern "C" int devias_grad_sumsq_multi(const devias_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                                       int32_t n_chunks, float* partials, void* stream) {
    DEVIAS_REQUIRE(table && chunk_tensor && chunk_index && partials && n_chunks >= 0, "devias_grad_sumsq_multi: bad args");
    if (n_chunks == 0) return DEVIAS_OK;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_index, partials);
    DEVIAS_CHECK_LAUNCH("devias_grad_sumsq_multi");
    return DEVIAS_OK;
}

extern "C" int devias_clip_coef(const float* partials, int32_t n_chunks, float max_norm, float* out, void* stream) {
    DEVIAS_REQUIRE(out && n_chunks >= 0 && (partials || n_chunks == 0), "devias_clip_coef: bad args");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n_chunks, max_norm, out);
    DEVIAS_CHECK_LAUNCH("devias_clip_coef");
    return DEVIAS_OK;
}

extern "C" int devias_adamw_multi(const devias_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                                  int32_t n_chunks, float beta1, float beta2, float eps, float grad_scale,
                                  const float* grad_scale_dev, void* stream) {
    DEVIAS_REQUIRE(table && chunk_tensor && chunk_index && n_chunks >= 0, "devias_adamw_multi: bad args");
    if (n_chunks == 0) return DEVIAS_OK;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_index, beta1, beta2,
                       eps, grad_scale, grad_scale_dev);
    DEVIAS_CHECK_LAUNCH("devias_adamw_multi");
    return DEVIAS_OK;
}


// ---- measurement aid: hold K compute units for a while (bench.py --cu-hog) -----------------------------------------------------------------
// K workgroups that each pin 128 KiB of LDS (so that no 128-KiB-LDS GEMM workgroup can share their CU) and spin for `usec` microseconds of the
// 100 MHz clock.  Launched on a side stream during backward it stands in for the CUs a concurrent RCCL kernel would occupy: how the persistent
// GEMM grids (one workgroup per CU, static tile lists) degrade with K CUs missing can be measured on ONE GPU.
__global__ __launch_bounds__(64) void cu_hog_kernel(unsigned long long ticks, int* sink) {
    __shared__ int pin[32768];
    pin[threadIdx.x] = (int)blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && pin[(threadIdx.x * 7) & 63] == -12345) sink[0] = 1;       // keeps the LDS array alive
}
extern "C" int devias_debug_cu_hog(int32_t n_workgroups, int32_t usec, void* stream) {
    DEVIAS_REQUIRE(n_workgroups >= 0 && n_workgroups <= 256 && usec >= 0 && usec <= 2000000, "devias_debug_cu_hog: bad args");
    if (n_workgroups == 0 || usec == 0) return DEVIAS_OK;
    hipLaunchKernelGGL(cu_hog_kernel, dim3(n_workgroups), dim3(64), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, (int*)nullptr);
    DEVIAS_CHECK_LAUNCH("devias_debug_cu_hog");
    return DEVIAS_OK;
}
