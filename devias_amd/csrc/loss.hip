// Slot selection + TrainLoss ('matching', scene criterion KL or CE) for gfx950: one workgroup per sample, everything on
// the device (the reference does B SciPy calls on the host plus six .item() syncs per step,
// utils/loss/train_loss.py:112-122,183-187).  All statistics in fp32; reductions are wave shuffles + a 4-wave LDS combine.
#include "common.h"

namespace {

enum { LMAXS = 8 };

__device__ __forceinline__ float block_sum(float v, float* sm /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__device__ __forceinline__ float block_min(float v, float* sm) { return -block_max(-v, sm); }

// argmax with first-occurrence tie-break (torch.argmax semantics on CPU): reduce (value, index) pairs
__device__ __forceinline__ void block_argmax(float v, int idx, float* smv, int* smi, float& ov, int& oi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float v2 = __shfl_xor(v, o, 64); int i2 = __shfl_xor(idx, o, 64);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { smv[threadIdx.x >> 6] = v; smi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    ov = smv[0]; oi = smi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (smv[w] > ov || (smv[w] == ov && smi[w] < oi)) { ov = smv[w]; oi = smi[w]; }
}

// row softmax statistics of Z[row, 0..C): max and logsumexp
template <typename T>
__device__ __forceinline__ void row_stats(const T* z, int C, float* sm, float& mx, float& lse) {
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, to_f32(z[c]));
    mx = block_max(m, sm);
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += expf(to_f32(z[c]) - mx);
    lse = mx + logf(block_sum(s, sm));
}

struct TeacherStats { float pad, lse; int argmax; };
__device__ __forceinline__ TeacherStats teacher_stats(const float* teacher, int B, int ns, int nb, int b, float* sm, int* smi) {
    TeacherStats r;
    float mn = INFINITY;
    for (int i = threadIdx.x; i < B * ns; i += 256) mn = fminf(mn, teacher[i]);
    r.pad = block_min(mn, sm) - 1.0f;                            // train_loss.py:103 (min over the rank-local batch)
    const float* t = teacher + (int64_t)b * ns;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < ns; c += 256) { float v = t[c]; if (v > bv) { bv = v; bi = c; } }
    float tmax; block_argmax(bv, bi, sm, smi, tmax, r.argmax);
    float s = 0.f;
    for (int c = threadIdx.x; c < ns; c += 256) s += expf(t[c] - tmax);
    s = block_sum(s, sm) + (float)nb * expf(r.pad - tmax);
    r.lse = tmax + logf(s);
    return r;
}

// ---- slot selection (modeling_slot.py:395-401) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void slot_select_kernel(const T* __restrict__ Z, int S, int C, int nb, int* __restrict__ idx) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    float best_a = -INFINITY, best_s = -INFINITY; int ia = 0, is = 0;
    for (int s = 0; s < S; ++s) {
        const T* z = Z + ((int64_t)b * S + s) * C;
        float mx, lse; row_stats(z, C, sm, mx, lse);
        float ma = -INFINITY, ms = -INFINITY;
        for (int c = threadIdx.x; c < C; c += 256) {
            float v = to_f32(z[c]);
            if (c < nb) ma = fmaxf(ma, v); else ms = fmaxf(ms, v);
        }
        ma = expf(block_max(ma, sm) - lse);      // max prob over the action / scene classes of slot s
        ms = expf(block_max(ms, sm) - lse);
        if (ma > best_a) { best_a = ma; ia = s; }   // strict > : first slot wins ties (torch.argmax)
        if (ms > best_s) { best_s = ms; is = s; }
    }
    if (threadIdx.x == 0) { idx[2 * b] = ia; idx[2 * b + 1] = is; }
}

// ---- loss forward ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void loss_fwd_kernel(devias_loss_dims d, const T* __restrict__ Z, const T* __restrict__ slots,
                                                       const T* __restrict__ maskp, const float* __restrict__ attn,
                                                       const float* __restrict__ teacher, const int64_t* __restrict__ target,
                                                       const float* __restrict__ fg, const float* __restrict__ fgN,
                                                       float* __restrict__ per_sample, int* __restrict__ match,
                                                       T* __restrict__ out_logits) {
    __shared__ float sm[4];
    __shared__ int smi[4];
    __shared__ float s_lse[LMAXS];
    __shared__ int s_ij[2];
    const int b = blockIdx.x, S = d.S, C = d.C;
    const int y = (int)target[b];
    TeacherStats ts = teacher_stats(teacher, d.B, d.ns, d.nb, b, sm, smi);
    const int st = d.nb + ts.argmax;                                           // train_loss.py:100,107
    for (int s = 0; s < S; ++s) {
        float mx, lse; row_stats(Z + ((int64_t)b * S + s) * C, C, sm, mx, lse);
        if (threadIdx.x == 0) s_lse[s] = lse;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // cost[s][0] = -p[s,y], cost[s][1] = -p[s,st]; argmin over ordered pairs i != j (== linear_sum_assignment, :112-122)
        float best = INFINITY; int bi = 0, bj = (S > 1 ? 1 : 0);
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                if (i == j) continue;
                const T* zi = Z + ((int64_t)b * S + i) * C; const T* zj = Z + ((int64_t)b * S + j) * C;
                float c = -expf(to_f32(zi[y]) - s_lse[i]) - expf(to_f32(zj[st]) - s_lse[j]);
                if (c < best) { best = c; bi = i; bj = j; }
            }
        s_ij[0] = bi; s_ij[1] = bj;
        match[2 * b] = bi; match[2 * b + 1] = bj;
    }
    __syncthreads();
    const int is = s_ij[0], js = s_ij[1];
    const T* zi = Z + ((int64_t)b * S + is) * C;
    const T* zj = Z + ((int64_t)b * S + js) * C;
    const float act = s_lse[is] - to_f32(zi[y]);                               // CE, :150
    // scene_criterion 'KL': KL(T || softmax(Z_j)) with 'batchmean' on a 1-D input => / C, times w_scene (:159-164);
    // 'CE': cross-entropy of slot j* against the teacher's argmax class, NOT weighted (:155-156)
    float kl = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float lt = (c < d.nb ? ts.pad : teacher[(int64_t)b * d.ns + c - d.nb]) - ts.lse;
        float lz = to_f32(zj[c]) - s_lse[js];
        kl += expf(lt) * (lt - lz);
        out_logits[(int64_t)b * C + c] = zi[c];
    }
    kl = d.scene_ce ? s_lse[js] - to_f32(zj[st]) : block_sum(kl, sm) * d.w_scene / (float)C;
    // BCE-with-logits on the already-sigmoided prediction (double sigmoid, :146-149)
    float mp = 0.f;
    const T* mrow = maskp + ((int64_t)b * S + is) * d.G;
    for (int k = threadIdx.x; k < d.G; k += 256) {
        float x = to_f32(mrow[k]), t = fg[(int64_t)b * d.G + k];
        mp += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    }
    mp = block_sum(mp, sm) * d.w_mask_pred / (float)d.G;
    // mask distillation: MSE(mean_h A[(b,h), i*, :], fgN[b])  (:97,145)
    float md = 0.f;
    for (int j = threadIdx.x; j < d.N; j += 256) {
        float a = 0.f;
        for (int hh = 0; hh < d.nh; ++hh) a += attn[(((int64_t)b * d.nh + hh) * S + is) * d.N + j];
        a = a / (float)d.nh - fgN[(int64_t)b * d.N + j];
        md += a * a;
    }
    md = block_sum(md, sm) * d.w_mask_distill / (float)d.N;
    // cosine loss over slot pairs (:173-178)
    float nrm[LMAXS];
    for (int s = 0; s < S; ++s) {
        float q = 0.f;
        const T* x = slots + ((int64_t)b * S + s) * d.D;
        for (int k = threadIdx.x; k < d.D; k += 256) { float v = to_f32(x[k]); q += v * v; }
        nrm[s] = fmaxf(sqrtf(block_sum(q, sm)), 1e-12f);
    }
    float cs = 0.f;
    for (int i = 0; i < S; ++i)
        for (int j = i + 1; j < S; ++j) {
            float q = 0.f;
            const T* xi = slots + ((int64_t)b * S + i) * d.D; const T* xj = slots + ((int64_t)b * S + j) * d.D;
            for (int k = threadIdx.x; k < d.D; k += 256) q += to_f32(xi[k]) * to_f32(xj[k]);
            cs += 2.0f * block_sum(q, sm) / (nrm[i] * nrm[j]);
        }
    cs = S > 1 ? cs / (float)(S * (S - 1)) : 0.f;
    if (threadIdx.x == 0) {
        float* o = per_sample + (int64_t)b * 5;
        o[0] = act; o[1] = kl; o[2] = cs; o[3] = mp; o[4] = md;
    }
}

__global__ void loss_final_kernel(const float* __restrict__ per_sample, int B, float* __restrict__ out) {
    if (threadIdx.x < 5) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += per_sample[(int64_t)b * 5 + threadIdx.x];
        out[threadIdx.x] = s / (float)B;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[5] = out[0] + out[1] + out[2] + out[3] + out[4];     // :180
}

// ---- loss backward -----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(devias_loss_dims d, const T* __restrict__ Z, const T* __restrict__ slots,
                                                       const T* __restrict__ maskp, const float* __restrict__ attn,
                                                       const float* __restrict__ teacher, const int64_t* __restrict__ target,
                                                       const float* __restrict__ fg, const float* __restrict__ fgN,
                                                       const int* __restrict__ match, const float* __restrict__ g_total,
                                                       T* __restrict__ dZ, T* __restrict__ dslots, T* __restrict__ dmaskp,
                                                       float* __restrict__ dattn) {
    __shared__ float sm[4];
    __shared__ int smi[4];
    const int b = blockIdx.x, S = d.S, C = d.C;
    const float g = g_total[0] / (float)d.B;
    const int y = (int)target[b];
    const int is = match[2 * b], js = match[2 * b + 1];
    TeacherStats ts = teacher_stats(teacher, d.B, d.ns, d.nb, b, sm, smi);
    float lse_i, lse_j, mx;
    row_stats(Z + ((int64_t)b * S + is) * C, C, sm, mx, lse_i);
    row_stats(Z + ((int64_t)b * S + js) * C, C, sm, mx, lse_j);
    const float wk = d.w_scene / (float)C;
    const int st = d.nb + ts.argmax;
    for (int s = 0; s < S; ++s) {
        const T* z = Z + ((int64_t)b * S + s) * C;
        T* dz = dZ + ((int64_t)b * S + s) * C;
        for (int c = threadIdx.x; c < C; c += 256) {
            float v = 0.f;
            if (s == is) v += g * (expf(to_f32(z[c]) - lse_i) - (c == y ? 1.f : 0.f));
            if (s == js) {
                if (d.scene_ce) v += g * (expf(to_f32(z[c]) - lse_j) - (c == st ? 1.f : 0.f));
                else {
                    float lt = (c < d.nb ? ts.pad : teacher[(int64_t)b * d.ns + c - d.nb]) - ts.lse;
                    v += g * wk * (expf(to_f32(z[c]) - lse_j) - expf(lt));
                }
            }
            dz[c] = from_f32<T>(v);
        }
        // mask prediction: d/dx of BCEwithLogits(x, t) = (sigmoid(x) - t) / G, x = the (already sigmoided) prediction
        const T* mrow = maskp + ((int64_t)b * S + s) * d.G;
        T* dm = dmaskp + ((int64_t)b * S + s) * d.G;
        for (int k = threadIdx.x; k < d.G; k += 256) {
            float v = 0.f;
            if (s == is) {
                float x = to_f32(mrow[k]), t = fg[(int64_t)b * d.G + k];
                v = g * d.w_mask_pred / (float)d.G * (1.0f / (1.0f + expf(-x)) - t);
            }
            dm[k] = from_f32<T>(v);
        }
        // mask distillation gradient on the returned attention (all heads share the head-mean)
        for (int j = threadIdx.x; j < d.N; j += 256) {
            float v = 0.f;
            if (s == is) {
                float a = 0.f;
                for (int hh = 0; hh < d.nh; ++hh) a += attn[(((int64_t)b * d.nh + hh) * S + s) * d.N + j];
                a = a / (float)d.nh - fgN[(int64_t)b * d.N + j];
                v = g * d.w_mask_distill * 2.0f / (float)d.N * a / (float)d.nh;
            }
            for (int hh = 0; hh < d.nh; ++hh) dattn[(((int64_t)b * d.nh + hh) * S + s) * d.N + j] = v;
        }
    }
    // cosine loss gradient: L = mean_b sum_{i != j} n_i.n_j / (S(S-1));  dL/dx_i = 2/(S(S-1)B) sum_{j != i} (n_j - (n_i.n_j) n_i)/|x_i|
    float nrm[LMAXS];
    float dots[LMAXS][LMAXS];
    for (int s = 0; s < S; ++s) {
        float q = 0.f;
        const T* x = slots + ((int64_t)b * S + s) * d.D;
        for (int k = threadIdx.x; k < d.D; k += 256) { float v = to_f32(x[k]); q += v * v; }
        nrm[s] = fmaxf(sqrtf(block_sum(q, sm)), 1e-12f);
    }
    for (int i = 0; i < S; ++i)
        for (int j = i + 1; j < S; ++j) {
            float q = 0.f;
            const T* xi = slots + ((int64_t)b * S + i) * d.D; const T* xj = slots + ((int64_t)b * S + j) * d.D;
            for (int k = threadIdx.x; k < d.D; k += 256) q += to_f32(xi[k]) * to_f32(xj[k]);
            float v = block_sum(q, sm) / (nrm[i] * nrm[j]);
            dots[i][j] = v; dots[j][i] = v;
        }
    const float gc = S > 1 ? g * 2.0f / (float)(S * (S - 1)) : 0.f;
    for (int i = 0; i < S; ++i) {
        const T* xi = slots + ((int64_t)b * S + i) * d.D;
        T* dxi = dslots + ((int64_t)b * S + i) * d.D;
        for (int k = threadIdx.x; k < d.D; k += 256) {
            float ni = to_f32(xi[k]) / nrm[i];
            float acc = 0.f;
            for (int j = 0; j < S; ++j) {
                if (j == i) continue;
                float nj = to_f32(slots[((int64_t)b * S + j) * d.D + k]) / nrm[j];
                acc += nj - dots[i][j] * ni;
            }
            dxi[k] = from_f32<T>(gc * acc / nrm[i]);
        }
    }
}

}  // namespace

extern "C" int devias_slot_select(const void* slots_head, int32_t dtype, int32_t B, int32_t S, int32_t C, int32_t nb,
                                  int32_t* idx, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    DEVIAS_REQUIRE(slots_head && idx && B > 0 && S > 0 && C > nb && nb > 0, "devias_slot_select: bad args");
    if (dtype == DEVIAS_BF16) hipLaunchKernelGGL((slot_select_kernel<bf16>), dim3(B), dim3(256), 0, st, (const bf16*)slots_head, S, C, nb, idx);
    else if (dtype == DEVIAS_F32) hipLaunchKernelGGL((slot_select_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)slots_head, S, C, nb, idx);
    else return devias_set_error(DEVIAS_EINVAL, "devias_slot_select: bad dtype %d", dtype);
    DEVIAS_CHECK_LAUNCH("devias_slot_select");
    return DEVIAS_OK;
}

extern "C" int64_t devias_head_match_loss_workspace_bytes(int32_t B) { return (int64_t)B * 5 * 4; }

static int check_dims(const devias_loss_dims* d, const char* who) {
    if (!d) return devias_set_error(DEVIAS_EINVAL, "%s: null dims", who);
    if (d->B <= 0 || d->S < 1 || d->S > LMAXS || d->C != d->nb + d->ns || d->D <= 0 || d->G <= 0 || d->N <= 0 || d->nh <= 0)
        return devias_set_error(DEVIAS_EINVAL, "%s: bad dims B=%d S=%d C=%d nb=%d ns=%d D=%d G=%d N=%d nh=%d", who, d->B, d->S, d->C,
                                d->nb, d->ns, d->D, d->G, d->N, d->nh);
    if (d->dtype != DEVIAS_BF16 && d->dtype != DEVIAS_F32) return devias_set_error(DEVIAS_EINVAL, "%s: bad dtype %d", who, d->dtype);
    return DEVIAS_OK;
}

extern "C" int devias_head_match_loss_fwd(const devias_loss_dims* d, const void* slots_head, const void* slots, const void* maskp,
                                          const float* attn, const float* teacher, const int64_t* target, const float* fg,
                                          const float* fgN, float* out_losses, int32_t* out_match, void* out_logits, float* ws,
                                          void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_dims(d, "devias_head_match_loss_fwd");
    if (rc) return rc;
    DEVIAS_REQUIRE(slots_head && slots && maskp && attn && teacher && target && fg && fgN && out_losses && out_match && out_logits && ws,
                   "devias_head_match_loss_fwd: null pointer");
    if (d->dtype == DEVIAS_BF16)
        hipLaunchKernelGGL((loss_fwd_kernel<bf16>), dim3(d->B), dim3(256), 0, st, *d, (const bf16*)slots_head, (const bf16*)slots,
                           (const bf16*)maskp, attn, teacher, target, fg, fgN, ws, out_match, (bf16*)out_logits);
    else
        hipLaunchKernelGGL((loss_fwd_kernel<float>), dim3(d->B), dim3(256), 0, st, *d, (const float*)slots_head, (const float*)slots,
                           (const float*)maskp, attn, teacher, target, fg, fgN, ws, out_match, (float*)out_logits);
    DEVIAS_CHECK_LAUNCH("devias_head_match_loss_fwd");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, ws, d->B, out_losses);
    DEVIAS_CHECK_LAUNCH("devias_head_match_loss_fwd(final)");
    return DEVIAS_OK;
}

extern "C" int devias_head_match_loss_bwd(const devias_loss_dims* d, const void* slots_head, const void* slots, const void* maskp,
                                          const float* attn, const float* teacher, const int64_t* target, const float* fg,
                                          const float* fgN, const int32_t* match, const float* g_total, void* d_slots_head,
                                          void* d_slots, void* d_maskp, float* d_attn, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_dims(d, "devias_head_match_loss_bwd");
    if (rc) return rc;
    DEVIAS_REQUIRE(slots_head && slots && maskp && attn && teacher && target && fg && fgN && match && g_total && d_slots_head &&
                   d_slots && d_maskp && d_attn, "devias_head_match_loss_bwd: null pointer");
    if (d->dtype == DEVIAS_BF16)
        hipLaunchKernelGGL((loss_bwd_kernel<bf16>), dim3(d->B), dim3(256), 0, st, *d, (const bf16*)slots_head, (const bf16*)slots,
                           (const bf16*)maskp, attn, teacher, target, fg, fgN, match, g_total, (bf16*)d_slots_head, (bf16*)d_slots,
                           (bf16*)d_maskp, d_attn);
    else
        hipLaunchKernelGGL((loss_bwd_kernel<float>), dim3(d->B), dim3(256), 0, st, *d, (const float*)slots_head, (const float*)slots,
                           (const float*)maskp, attn, teacher, target, fg, fgN, match, g_total, (float*)d_slots_head, (float*)d_slots,
                           (float*)d_maskp, d_attn);
    DEVIAS_CHECK_LAUNCH("devias_head_match_loss_bwd");
    return DEVIAS_OK;
}
