// FAME foreground masks and clip mixing on the device (reference: utils/transform/fame.py; CPU + kornia there).
// HBM-bound image work on [B, 3, T, H, W] fp32 clips: one pass over the clip produces every frame-difference map and the HSV
// colour-bin map; masks are 11-tap separable Gaussians, per-image top-k selections (exact k-th value by a 3-pass radix select
// over the float bits, ties taken in index order -> deterministic), 1000-bin colour histograms in LDS with integer atomics, and a
// final select-by-mask mix of two clips.  No floating-point atomics anywhere.
#include "common.h"

namespace {

struct GaussTaps { float w[32]; };

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ---- K1: denormalise, frame differences, colour-bin map ---------------------------------------------------------------------
// diffs[b][0] = mean_t<T-1 sum_c |v_t - v_t+1| (fame.py:93); diffs[b][1 + i] = sum_c |v_2i - v_2i+1| (fame.py:107);
// cmap[b][p] = h + (s-1)*10 + (v-1)*100 from the HSV of the temporal mean image (fame.py:47-64; kornia rgb_to_hsv, hue in [0, 2pi])
__global__ __launch_bounds__(256) void fame_diff_color_kernel(const float* __restrict__ video, int B, int T, int HW,
                                                              float* __restrict__ diffs, short* __restrict__ cmap) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= HW) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int S = 1 + T / 2;
    const float* vb = video + (int64_t)b * 3 * T * HW + p;
    float prev[3], sum[3] = {0.f, 0.f, 0.f};
    float dsum = 0.f;
    float* db = diffs + (int64_t)b * S * HW + p;
    for (int t = 0; t < T; ++t) {
        float cur[3], d = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cur[c] = __fadd_rn(__fmul_rn(vb[((int64_t)c * T + t) * HW], stdv[c]), mean[c]);      // x * std + mean, two roundings
            sum[c] += cur[c];
            if (t > 0) d += fabsf(prev[c] - cur[c]);
            prev[c] = cur[c];
        }
        if (t > 0) {
            dsum += d;
            if (t & 1) db[(int64_t)(1 + (t >> 1)) * HW] = d;
        }
    }
    db[0] = dsum / (float)(T - 1);
    // kornia.color.rgb_to_hsv on the temporal mean
    const float r = sum[0] / (float)T, g = sum[1] / (float)T, bl = sum[2] / (float)T;
    float mx = r; int am = 0;
    if (g > mx) { mx = g; am = 1; }
    if (bl > mx) { mx = bl; am = 2; }
    const float mn = fminf(r, fminf(g, bl));
    float delta = mx - mn;
    const float v = mx, s = __fdiv_rn(delta, mx + 1e-8f);
    if (delta == 0.f) delta = 1.f;
    const float rc = mx - r, gc = mx - g, bc = mx - bl;
    float h = am == 0 ? (bc - gc) : (am == 1 ? (rc - bc) + 2.0f * delta : (gc - rc) + 4.0f * delta);
    h = __fdiv_rn(h, delta);
    h = __fdiv_rn(h, 6.0f);
    h = h - floorf(h);                                             // python-style (h / 6) % 1
    h = 6.283185307179586f * h;
    const float ang = h * 6.283185307179586f;                     // the reference multiplies by 2*pi once more (fame.py:57-58)
    const float hx = (s * cosf(ang) + 1.f) * 0.5f, hy = (s * sinf(ang) + 1.f) * 0.5f;
    const int hb = (int)rintf(hx * 9.f + 1.f), sb = (int)rintf(hy * 9.f + 1.f), vbn = (int)rintf(v * 9.f + 1.f);
    int bin = hb + (sb - 1) * 10 + (vbn - 1) * 100;
    bin = bin < 0 ? 0 : (bin > 999 ? 999 : bin);                  // bin 1000 (h = s = v = 10) overruns the reference's 1000-entry table
    cmap[(int64_t)b * HW + p] = (short)bin;
}

// ---- K2: separable Gaussian, 'reflect' border (kornia.filters.GaussianBlur2d) ---------------------------------------------
enum { BT = 32, BMAXR = 15 };
__global__ __launch_bounds__(256) void fame_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int R, GaussTaps g) {
    __shared__ float tile[BT + 2 * BMAXR][BT + 2 * BMAXR + 1];
    __shared__ float rowp[BT + 2 * BMAXR][BT + 1];
    const int img = blockIdx.z, x0 = blockIdx.x * BT, y0 = blockIdx.y * BT;
    const float* src = in + (int64_t)img * H * W;
    const int E = BT + 2 * R;
    for (int i = threadIdx.x; i < E * E; i += 256) {
        const int ty = i / E, tx = i % E;
        tile[ty][tx] = src[(int64_t)reflect(min(y0 + ty - R, H + R - 1), H) * W + reflect(min(x0 + tx - R, W + R - 1), W)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E * BT; i += 256) {              // along x
        const int ty = i / BT, tx = i % BT;
        float a = 0.f;
        for (int k = 0; k <= 2 * R; ++k) a += g.w[k] * tile[ty][tx + k];
        rowp[ty][tx] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BT * BT; i += 256) {             // along y
        const int ty = i / BT, tx = i % BT;
        if (y0 + ty >= H || x0 + tx >= W) continue;
        float a = 0.f;
        for (int k = 0; k <= 2 * R; ++k) a += g.w[k] * rowp[ty + k][tx];
        out[(int64_t)img * H * W + (int64_t)(y0 + ty) * W + x0 + tx] = a;
    }
}

// ---- exact k-th value of one image by radix select (block of 1024 threads) -----------------------------------------------------
enum { SEL_NT = 1024 };
__device__ __forceinline__ unsigned to_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);             // monotonic: larger float <=> larger key
}

struct SelShared {
    unsigned hist[2048];
    unsigned cnt[SEL_NT];
    unsigned prefix, mask; int remaining; int sel;
};

// after the call: key = the k-th best key; need = how many pixels EQUAL to it belong to the selection (>= 1)
template <bool LARGEST>
__device__ void radix_select(const float* __restrict__ x, int n, int k, SelShared& sh, unsigned& key, int& need) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { sh.prefix = 0; sh.mask = 0; sh.remaining = k; }
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = 1 << bits[pass];
        for (int i = tid; i < 2048; i += SEL_NT) sh.hist[i] = 0;
        __syncthreads();
        const unsigned prefix = sh.prefix, mask = sh.mask;
        for (int i = tid; i < n; i += SEL_NT) {
            const unsigned kk = to_key(x[i]);
            if ((kk & mask) == prefix) atomicAdd(&sh.hist[(kk >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                            // wave 0: find the digit whose cumulative count reaches `remaining`
            const int per = nb / 64;
            unsigned mine = 0;
            for (int j = 0; j < per; ++j) {
                const int bin = LARGEST ? nb - 1 - (lane * per + j) : lane * per + j;
                mine += sh.hist[bin];
            }
            unsigned incl = mine;                                  // inclusive prefix over lanes (lane 0 = best digits)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            const unsigned excl = incl - mine;
            const unsigned rem = (unsigned)sh.remaining;
            if (excl < rem && rem <= incl) {
                unsigned cum = excl;
                for (int j = 0; j < per; ++j) {
                    const int bin = LARGEST ? nb - 1 - (lane * per + j) : lane * per + j;
                    const unsigned h = sh.hist[bin];
                    if (cum + h >= rem) { sh.sel = bin; sh.remaining = (int)(rem - cum); break; }
                    cum += h;
                }
            }
        }
        __syncthreads();
        if (tid == 0) { sh.prefix |= (unsigned)sh.sel << shift; sh.mask |= (unsigned)(nb - 1) << shift; }
        __syncthreads();
    }
    key = sh.prefix; need = sh.remaining;
    __syncthreads();
}

// visit every selected pixel: keys strictly better than `key`, and the first `need` pixels (in index order) equal to it
template <bool LARGEST, typename F>
__device__ void for_each_selected(const float* __restrict__ x, int n, unsigned key, int need, SelShared& sh, F f) {
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < n; i += SEL_NT) {
        const unsigned kk = to_key(x[i]);
        if (LARGEST ? kk > key : kk < key) f(i);
    }
    const int chunk = (n + SEL_NT - 1) / SEL_NT, lo = tid * chunk, hi = min(n, lo + chunk);
    unsigned mine = 0;
    for (int i = lo; i < hi; ++i) mine += to_key(x[i]) == key;
    sh.cnt[tid] = mine;
    __syncthreads();
    if (tid < 64) {                                                // exclusive scan of the 1024 counts, 16 per lane
        unsigned s = 0;
        for (int j = 0; j < 16; ++j) s += sh.cnt[lane * 16 + j];
        unsigned incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        unsigned run = incl - s;
        for (int j = 0; j < 16; ++j) { const unsigned c = sh.cnt[lane * 16 + j]; sh.cnt[lane * 16 + j] = run; run += c; }
    }
    __syncthreads();
    int before = (int)sh.cnt[tid];
    for (int i = lo; i < hi && before < need; ++i)
        if (to_key(x[i]) == key) { f(i); ++before; }
    __syncthreads();
}

// ---- K3: colour model of one mask image -> refined soft mask (fame.py:50-76) ------------------------------------------------
__global__ __launch_bounds__(SEL_NT) void fame_seg_refine_kernel(const float* __restrict__ blurred, const short* __restrict__ cmap,
                                                                 int imgs_per_clip, int HW, int k_fg, int k_bg, float eps,
                                                                 float* __restrict__ refine) {
    __shared__ SelShared sh;
    __shared__ unsigned hfg[1000], hbg[1000];
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* x = blurred + (int64_t)img * HW;
    const short* cm = cmap + (int64_t)(img / imgs_per_clip) * HW;
    for (int i = tid; i < 1000; i += SEL_NT) { hfg[i] = 0; hbg[i] = 0; }
    unsigned key; int need;
    radix_select<true>(x, HW, k_fg, sh, key, need);
    for_each_selected<true>(x, HW, key, need, sh, [&](int i) { atomicAdd(&hfg[cm[i]], 1u); });
    radix_select<false>(x, HW, k_bg, sh, key, need);
    for_each_selected<false>(x, HW, key, need, sh, [&](int i) { atomicAdd(&hbg[cm[i]], 1u); });
    __syncthreads();
    const float sfg = (float)k_fg + eps, sbg = (float)(k_bg + 1000) + eps;
    float* out = refine + (int64_t)img * HW;
    for (int i = tid; i < HW; i += SEL_NT) {
        const int c = cm[i];
        const float pf = __fdiv_rn((float)hfg[c], sfg), pb = __fdiv_rn((float)hbg[c] + 1.0f, sbg);
        out[i] = __fdiv_rn(pf, pb + pf);
    }
}

// ---- K4: top-k binarisation + 16x16 average pooling (fame.py:78-87, 142-148) --------------------------------------------------
__global__ __launch_bounds__(SEL_NT) void fame_binarize_pool_kernel(const float* __restrict__ blurred, int H, int W, int num_fg, int pool,
                                                                    unsigned char* __restrict__ binmask, float* __restrict__ pooled) {
    __shared__ SelShared sh;
    __shared__ unsigned cell[1024];
    const int img = blockIdx.x, tid = threadIdx.x, HW = H * W;
    const int pw = W / pool, ncell = (H / pool) * pw;
    const float* x = blurred + (int64_t)img * HW;
    for (int i = tid; i < ncell; i += SEL_NT) cell[i] = 0;
    if (binmask) for (int i = tid; i < HW; i += SEL_NT) binmask[(int64_t)img * HW + i] = 0;
    unsigned key; int need;
    radix_select<true>(x, HW, num_fg, sh, key, need);
    for_each_selected<true>(x, HW, key, need, sh, [&](int i) {
        const int y = i / W, xx = i % W;
        if (y / pool < H / pool && xx / pool < pw) atomicAdd(&cell[(y / pool) * pw + xx / pool], 1u);
        if (binmask) binmask[(int64_t)img * HW + i] = 1;
    });
    __syncthreads();
    for (int i = tid; i < ncell; i += SEL_NT) pooled[(int64_t)img * ncell + i] = (float)cell[i] / (float)(pool * pool);
}

// ---- K5: out[j] = aug[j] ? select(mask[src[j]], video[src[j]], video[partner[j]]) : video[src[j]] (fame.py:124-138) --------------
__global__ __launch_bounds__(256) void fame_mix_kernel(const float* __restrict__ video, const unsigned char* __restrict__ binmask, int mask_stride,
                                                       const int* __restrict__ src, const int* __restrict__ partner, const int* __restrict__ aug,
                                                       float* __restrict__ out, int CT, int HW) {
    const int j = blockIdx.z, ct = blockIdx.y;
    const int s = src[j], pr = partner[j], a = aug[j];
    const float* vs = video + ((int64_t)s * CT + ct) * HW;
    const float* vp = video + ((int64_t)pr * CT + ct) * HW;
    const unsigned char* m = binmask + (int64_t)s * mask_stride;
    float* o = out + ((int64_t)j * CT + ct) * HW;
    if ((HW & 3) == 0) {
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(vs + i);
            if (a) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(vp + i);
                const uchar4 mm = *reinterpret_cast<const uchar4*>(m + i);
                v[0] = mm.x ? v[0] : w[0]; v[1] = mm.y ? v[1] : w[1]; v[2] = mm.z ? v[2] : w[2]; v[3] = mm.w ? v[3] : w[3];
            }
            *reinterpret_cast<f32x4*>(o + i) = v;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) o[i] = (a && !m[i]) ? vp[i] : vs[i];
    }
}

}  // namespace

extern "C" int devias_fame_diff_color(const float* video, int32_t B, int32_t T, int32_t H, int32_t W, float* diffs, int16_t* cmap, void* stream) {
    DEVIAS_REQUIRE(video && diffs && cmap && B > 0 && T >= 2 && (T % 2) == 0 && H > 0 && W > 0, "devias_fame_diff_color: bad args (T must be even)");
    const int HW = H * W;
    hipLaunchKernelGGL(fame_diff_color_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, video, B, T, HW, diffs, cmap);
    DEVIAS_CHECK_LAUNCH("devias_fame_diff_color");
    return DEVIAS_OK;
}

extern "C" int devias_fame_blur(const float* in, float* out, int32_t n_img, int32_t H, int32_t W, int32_t ksize, float sigma, void* stream) {
    DEVIAS_REQUIRE(in && out && in != out && n_img > 0 && (ksize & 1) && ksize >= 1 && ksize <= 2 * BMAXR + 1 && ksize / 2 < H && ksize / 2 < W && sigma > 0.f,
                   "devias_fame_blur: odd kernel size <= 31, smaller than the image, distinct buffers");
    GaussTaps g;                                                   // kornia: exp(-x^2 / (2 sigma^2)) normalised, fp32
    float sum = 0.f;
    for (int i = 0; i < 32; ++i) g.w[i] = 0.f;
    for (int i = 0; i < ksize; ++i) { const float x = (float)(i - ksize / 2); g.w[i] = expf(-(x * x) / (2.0f * sigma * sigma)); sum += g.w[i]; }
    for (int i = 0; i < ksize; ++i) g.w[i] /= sum;
    hipLaunchKernelGGL(fame_blur_kernel, dim3((W + BT - 1) / BT, (H + BT - 1) / BT, n_img), dim3(256), 0, (hipStream_t)stream, in, out, H, W, ksize / 2, g);
    DEVIAS_CHECK_LAUNCH("devias_fame_blur");
    return DEVIAS_OK;
}

extern "C" int devias_fame_seg_refine(const float* blurred, const int16_t* cmap, int32_t n_img, int32_t imgs_per_clip, int32_t HW, float eps,
                                      float* refine, void* stream) {
    DEVIAS_REQUIRE(blurred && cmap && refine && n_img > 0 && imgs_per_clip > 0 && HW >= 10, "devias_fame_seg_refine: bad args");
    const int k_fg = (int)(0.5 * HW), k_bg = (int)(0.1 * HW);
    hipLaunchKernelGGL(fame_seg_refine_kernel, dim3(n_img), dim3(SEL_NT), 0, (hipStream_t)stream, blurred, cmap, imgs_per_clip, HW, k_fg, k_bg, eps, refine);
    DEVIAS_CHECK_LAUNCH("devias_fame_seg_refine");
    return DEVIAS_OK;
}

extern "C" int devias_fame_binarize_pool(const float* blurred, int32_t n_img, int32_t H, int32_t W, int32_t num_fg, int32_t pool,
                                         uint8_t* binmask, float* pooled, void* stream) {
    DEVIAS_REQUIRE(blurred && pooled && n_img > 0 && pool > 0 && (H / pool) * (W / pool) <= 1024 && num_fg >= 1 && num_fg <= H * W,
                   "devias_fame_binarize_pool: bad args");
    hipLaunchKernelGGL(fame_binarize_pool_kernel, dim3(n_img), dim3(SEL_NT), 0, (hipStream_t)stream, blurred, H, W, num_fg, pool, binmask, pooled);
    DEVIAS_CHECK_LAUNCH("devias_fame_binarize_pool");
    return DEVIAS_OK;
}

extern "C" int devias_fame_mix(const float* video, const uint8_t* binmask, int32_t mask_stride, const int32_t* src, const int32_t* partner,
                               const int32_t* aug, float* out, int32_t B, int32_t CT, int32_t HW, void* stream) {
    DEVIAS_REQUIRE(video && binmask && src && partner && aug && out && video != out && B > 0 && CT > 0 && HW > 0, "devias_fame_mix: bad args");
    int gx = (HW / 4 + 255) / 256; if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(fame_mix_kernel, dim3(gx, CT, B), dim3(256), 0, (hipStream_t)stream, video, binmask, mask_stride, src, partner, aug, out, CT, HW);
    DEVIAS_CHECK_LAUNCH("devias_fame_mix");
    return DEVIAS_OK;
}
