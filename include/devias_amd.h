/*
 * devias_amd.h -- C ABI of libdevias_amd.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * DEVIAS slot-ViT training step.
 *
 * The reference (KHU-VLL/DEVIAS) is pure Python on PyTorch: it has no FFI of its own.  The "binding" for
 * this path is therefore the set of ATen ops its modules dispatch to; every entry point below names the
 * reference code (file:line, relative to the reference root) whose arithmetic it replaces.  The Python
 * host (the devias_amd Python package) mirrors the reference's timm-style module surface and reaches these symbols
 * through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / STL types cross the boundary.
 *   - every pointer is DEVICE memory owned by the caller (PyTorch's caching allocator), including
 *     workspaces; the library allocates nothing.
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no device sync).
 *   - return 0 on success, <0 on error (DEVIAS_E*); devias_last_error() gives a thread-local message.
 *   - dtype codes: DEVIAS_F32 = 0, DEVIAS_BF16 = 1.  "T" below means the activation dtype of the call.
 *     Accumulation, softmax / LayerNorm statistics, losses and weight gradients are always fp32.
 *   - one process per GPU (as the reference: torchrun, utils/utils.py:251-277); calls are re-entrant.
 */
#ifndef DEVIAS_AMD_H
#define DEVIAS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEVIAS_OK 0
#define DEVIAS_EINVAL (-1)
#define DEVIAS_ELAUNCH (-2)
#define DEVIAS_EUNSUPPORTED (-3)

#define DEVIAS_F32 0
#define DEVIAS_BF16 1

/* epilogue activation codes for devias_gemm */
#define DEVIAS_ACT_NONE 0
#define DEVIAS_ACT_GELU 1      /* exact erf GELU (nn.GELU(), modeling_slot.py:55,63); pre-activation -> aux_out */
#define DEVIAS_ACT_RELU 2      /* MaskPredictor ReLU, modeling_slot.py:200-202 */
#define DEVIAS_ACT_SIGMOID 3   /* MaskPredictor Sigmoid, modeling_slot.py:204 */
#define DEVIAS_ACT_DGELU 4     /* backward: v *= gelu'(aux_in) with aux_in = saved pre-activation */
#define DEVIAS_ACT_DRELU 5     /* backward: v = aux_in > 0 ? v : 0 with aux_in = saved ReLU output */

int devias_version(void);          /* 100 + additions: 110 = multi-tensor optimizer entry points, 120 = devias_fame_*, 130 = counters + options,
                                    140 = (stream-K GEMM schedule, retired in 160; devias_gemm_args grew by sk_ws / sk_ws_bytes: recompile callers), devias_allreduce_bucket,
                                    150 = fused regions (devias_encoder_block_* / devias_agg_block_* / devias_head_*), devias_range_* */
const char* devias_last_error(void);
/* Launch counters: one per kernel family, incremented by the host side of each entry point (process-wide, relaxed atomics).
 * Tests use them to ASSERT that the kernels a parity claim is made for are the kernels that ran (the reference has no analogue:
 * its dispatch is ATen's). */
#define DEVIAS_CNT_GEMM128_F32 0     /* 128x128 register-staged kernel, exact fp32 MFMA */
#define DEVIAS_CNT_GEMM128_BF16 1    /* 128x128 register-staged kernel, bf16 (ragged / small shapes) */
#define DEVIAS_CNT_GEMM_SS 2         /* 256x128 single-stage LDS-DMA kernel */
#define DEVIAS_CNT_GEMM256 3         /* 256x256 two-stage LDS-DMA kernel, one tile per workgroup (wgrad / split-K / <= 1 round) */
#define DEVIAS_CNT_GEMM256P 4        /* 256x256 persistent kernel (forward / dgrad GEMMs of the measured step) */
#define DEVIAS_CNT_SPLITK_REDUCE 5
#define DEVIAS_CNT_MHSA_FWD_BF16 6   /* MFMA flash forward */
#define DEVIAS_CNT_MHSA_BWD_BF16 7   /* MFMA backward, two kernels (dQ; dK/dV) */
#define DEVIAS_CNT_MHSA_FWD_F32 8    /* VALU parity kernels */
#define DEVIAS_CNT_MHSA_BWD_F32 9
#define DEVIAS_CNT_GEMM_SK 11        /* (retired with ABI 160: always 0) */
#define DEVIAS_CNT_GEMM_SMALLM 13     /* small-M kernel (M <= 128, bf16, B k-contiguous): one launch instead of split-K product + reduce */
#define DEVIAS_CNT_GEMM256W 12       /* 256x256 persistent kernel, four-wave form (one wave per SIMD, accumulators in AGPRs; option gemm_w4, off by
                                        default); every such launch also counts as DEVIAS_CNT_GEMM256P */
#define DEVIAS_CNT_GEMM256D 14       /* 256x256 persistent kernel pulling its tiles from the per-XCD dynamic queues (option gemm_dynamic, on by default);
                                        every such launch also counts as DEVIAS_CNT_GEMM256P */
#define DEVIAS_CNT_MHSA_BWD_FUSED 10 /* (retired with ABI 150: always 0) */
/* which dK / dV kernel served a bf16 attention backward (ABI 165; each devias_mhsa_bwd* call also counts DEVIAS_CNT_MHSA_BWD_BF16) */
#define DEVIAS_CNT_DKDV1W 15         /* one wave per SIMD (csrc/attn_bwd1w.hip), one workgroup per 256-key block: the default of the measured step */
#define DEVIAS_CNT_DKDV1W_PERS 16    /* the same kernel as one persistent workgroup per CU (option attn_dkdv = 2) */
#define DEVIAS_CNT_DKDV1W_REST 17    /* its second launch for the N mod 256 last keys of every head (32 keys at N = 1568) */
#define DEVIAS_CNT_DKDV2W 18         /* the two-waves-per-SIMD kernel of rounds 2-4 (attention dropout, ws = NULL, option attn_dkdv = 0) */
#define DEVIAS_CNT_MHSA_QPRE 19      /* bf16 attention calls (forward or backward) served with DEVIAS_ATTN_Q_PRESCALED (ABI 167): 2 per encoder block and step in the measured path */
#define DEVIAS_CNT_MAX 24
int64_t devias_counter(int32_t id);          /* -1 for an unknown id */
void devias_counters_reset(void);
/* Process-wide integer options (initialised once from the DEVIAS_* environment variables of the same meaning): "gemm_epi",
 * "gemm256", "gemm_ss", "gemm_groupm", "gemm_persistent", "gemm_tail_split" (the tiles of a persistent launch's last partial round are computed in pieces by several
 * workgroups when enough CUs would idle: 3, default: as thirds where three pieces per tail tile fit the idle workgroups, else 128-row halves; 4: quarters / thirds / halves
 * (thirds and quarters on static tile lists, B k-contiguous, no column sums); 2: halves, their idle waves staging no A rows; 1: halves, staging all rows; 0: whole tiles), "gemm_smallm" (1, default: bf16 products with M <= 128 and B k-contiguous run as ONE launch of the small-M kernel instead of split-K
 * product + reduce; same epilogue arithmetic, different K summation order), "gemm_w4" (mask of the forms the four-wave persistent kernel serves; -1, default: the measured policy -- none since round 6's work on the eight-wave kernel; rounds 4-5: all four where K >= 1024 and N >= 1024, i.e. every GEMM of ViT-L and none of ViT-B; 15 = all four), "gemm_debug", "gemm_dynamic" (1: the workgroups of
 * the persistent kernel pull their tiles from per-XCD queues at run time instead of walking static lists -- a CU held or slowed by a concurrent kernel, e.g. RCCL's during
 * backward, just takes fewer tiles; 0: static lists; -1, default: queues exactly when "gemm_concurrent" is set; same bits either way), "gemm_concurrent" (the host
 * announces that other kernels run beside the step's: devias_amd.parallel.GradSync sets it for N > 1), "gemm_reserve_cus" (CUs the big-tile grids and the weight-gradient split-K sizing leave free for such a kernel),
 * "gemm_splitk_xcd" (1, default: a split-K launch of the 256 x 256 kernel -- the weight gradients -- orders its (slab, tile) pairs XCD-major, ~32 tiles of ONE slab per
 * XCD, so that an L2 holds one K range instead of seven: 2.3 x fewer operand bytes from the fabric; 0: the (tile, slab) grid; same bits),
 * "attn_cfg", "attn_xcd", "attn_bias_fused" (1, default: devias_mhsa_bwd_bias takes the q_bias / v_bias gradients from the backward kernels' accumulators; 0: by two
 * column-sum passes over dqkv, as before ABI 162 -- A/B aid), "regions_defer" (1, default: an encoder block's backward runs the second stages of its partial reductions -- LayerNorm parameter
 * gradients, bias-gradient column sums -- as ONE launch at its end instead of 5-7).  Every choice computes the same bits EXCEPT gemm_smallm (different K summation order), attn_bias_fused (the two bias gradients are sums of the kernels' fp32 accumulators instead of the bf16-rounded tensor) and gemm_reserve_cus (like the device's CU count it
 * sets the split-K factor of the weight-gradient GEMMs, hence their fp32 summation order: runs with different reserves -- or N = 1 against N > 1 runs that set one --
 * agree to rounding, not bitwise).  0 = ok, DEVIAS_EINVAL = unknown name. */
int devias_set_option(const char* name, int32_t value);
/* The current value of one of those options (so that a host can restore what it changed: devias_amd.parallel.GradSync.remove()). */
int devias_get_option(const char* name, int32_t* value);
/* The dynamic tile queues ("gemm_dynamic") live in ONE ring of queue slots per device, whose protocol (launch n zeroes the slot launch n + 32 will use)
 * relies on the launches being ordered among themselves.  The ring therefore belongs to one stream per device: the first stream that launches a
 * dynamic-queue GEMM on the device.  Launches on any other stream of that device silently walk the static tile lists (same tiles, same bits).  A host that
 * moves its step to another stream calls this once every dynamic-queue launch of the old stream has COMPLETED (e.g. after a device synchronise): the ring
 * is zeroed on `stream`, which becomes its owner.  One GPU per process or several: each device has its own ring and counter. */
int devias_gemm_release_queue_stream(void* stream);

/* In-place SUM all-reduce of one flat gradient bucket over the caller's RCCL communicator (`nccl_comm` is an ncclComm_t; dtype DEVIAS_F32 or
 * DEVIAS_BF16), enqueued on `stream`: the data-path collective of the step (DDP / DeepSpeed ZeRO-0 gradient all-reduce, run_slot_finetuning.py:552-563)
 * for hosts that own a communicator.  (The Python host in this repository uses torch.distributed -- the same RCCL -- see devias_amd/parallel.py.)
 * librccl is resolved at the first call; DEVIAS_EUNSUPPORTED if it cannot be found. */
int devias_allreduce_bucket(void* nccl_comm, void* bucket, int64_t count, int32_t dtype, void* stream);
/* The library allocates nothing persistent; resets the launch counters. */
void devias_shutdown(void);
/* ROCTX ranges (`rocprofv3 --marker-trace`): the fused regions open one each; hosts mark their own phases (loss, optimizer, step) with these.
 * Active only when the environment variable DEVIAS_ROCTX is non-zero (the marker library is then resolved with dlopen); otherwise no-ops.
 * The reference has no profiler ranges (utils/utils.py:120-164 keeps wall-clock meters only). */
void devias_range_push(const char* name);
void devias_range_pop(void);
/* ---------------------------------------------------------------------------------------------------
 * MEASUREMENT entry points (devias_debug_*).  Part of the ABI: bench.py's `roofline` object is built from three of them
 * (devias_debug_gemm_timer_arm / _read: in-step kernel timing; devias_debug_mfma_probe: the sustained MFMA peak of the box), its
 * --cu-hog flag from a fourth.  None of them changes what any other entry point computes; none has a reference counterpart.
 * ------------------------------------------------------------------------------------------------- */
/* Measurement aid (bench.py --cu-hog K): n_workgroups workgroups that each pin 128 KiB of LDS -- no 128-KiB-LDS GEMM workgroup can share their
 * CU -- and idle for `usec` microseconds, enqueued on `stream` (a side stream): a stand-in for the compute units a concurrent RCCL kernel
 * occupies during backward, so that the persistent GEMM grids' sensitivity to missing CUs can be measured on one GPU.  No reference analogue. */
int devias_debug_cu_hog(int32_t n_workgroups, int32_t usec, void* stream);
/* Measurement aid (bench.py's `roofline.sustained_peak`): a bare bf16 MFMA loop -- `n_workgroups` workgroups of 8 waves, operands in registers loaded once from
 * `data_bf16` (random values; >= 6144 elements, 16-byte aligned), `iters` iterations of 64 v_mfma_f32_16x16x32_bf16 per wave.  `sink` ([n_workgroups * 512] floats)
 * keeps the accumulators alive; `stamps` (optional, [n_workgroups][2] uint64) receives per workgroup the shader-clock cycles and the 100 MHz ticks its loop took:
 * their quotient x 100 MHz is the clock the CU held.  devias_debug_mfma_probe_flops = the flops one launch executes.  No reference counterpart. */
int devias_debug_mfma_probe(const void* data_bf16, int64_t data_elems, int32_t n_workgroups, int32_t iters, float* sink, uint64_t* stamps, void* stream);
int64_t devias_debug_mfma_probe_flops(int32_t n_workgroups, int32_t iters);
/* Measurement aid (bench.py's `roofline.dominant_kernel`): HIP events around every devias_gemm of ONE shape, wherever it is issued from -- a fused region in the
 * middle of a real step --, so that a kernel is timed as it runs IN the step.  arm(M, N, K, trans_a, trans_b) starts collecting (M <= 0: stops and clears); at most
 * 64 launches are kept; read() waits for them and returns their number and summed duration (product + the split-K reduce / column-sum second stage that devias_gemm
 * launches behind it).  The dims are devias_gemm_args' M, N, K. */
int devias_debug_gemm_timer_arm(int32_t M, int32_t N, int32_t K, int32_t trans_a, int32_t trans_b);
int devias_debug_gemm_timer_read(int32_t* count, float* total_ms);
/* the same launches one by one, in launch order (ABI 166): each_ms[0 .. min(count, cap) - 1].  bench.py's `roofline.gemm_shapes` arms a signature for ONE step: where a forward
 * GEMM and a dgrad GEMM on a transposed weight copy share (M, N, K, layouts) -- fc1 and dfc2, proj and dproj, fc2 and dfc1 -- the step's first launches are the forward's. */
int devias_debug_gemm_timer_read_each(int32_t* count, float* each_ms, int32_t cap);
/* Diagnostic builds (-DDKDV_STAMP on csrc/attn_bwd1w.hip) only, DEVIAS_EUNSUPPORTED otherwise: shader-clock stamps of the one-wave-per-SIMD dK / dV kernel's last
 * launch -- per workgroup (the first n <= 4096) eight stamps (csrc/attn_bwd1w.hip: entry, loop entry, loop exit, exit, and four points of prologue / epilogue) -- copied
 * to host memory out[n][8]. */
int devias_debug_dkdv_stamps(uint64_t* out, int32_t n);

/* fills: [0]=CU count, [1]=max clock kHz, [2]=LDS bytes per block, [3]=wavefront size, [4]=gfx arch number (e.g. 950) */
int devias_device_info(int device, int64_t* out5);

/* ---------------------------------------------------------------------------------------------------
 * Generic fused GEMM:  C[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )
 *   trans_a = 0: A is [M,K] row-major (lda = row stride);  1: A is stored [K,M] (reduction-major)
 *   trans_b = 0: B is [N,K] row-major -- the nn.Linear weight layout;  1: B is stored [K,N]
 *   epilogue order: +bias[n] -> act (GELU stores the pre-activation to aux_out first) -> *row_scale[m / rows_per_scale] -> +res[(m % res_mod or m), n]
 *   c_f32 = 1 with T = bf16 writes C (and reads it under beta) as fp32: weight-gradient GEMMs.
 *   split_k > 1: partial sums go to `ws` (split_k * M * N floats) and a second kernel reduces them in a fixed order and
 *     applies the same epilogue (used for the long-M weight-gradient reductions and the 64-row slot MLP GEMMs).
 * Naming: SURVEY.md section 8(b) sketched this boundary as devias_gemm_bias_act_{fwd,dgrad,wgrad} and devias_patch_embed_{fwd,wgrad}.  They are ONE entry point here:
 *   devias_gemm_bias_act_fwd   = devias_gemm(trans_a 0, trans_b 0, bias, act, aux_out, res)
 *   devias_gemm_bias_act_dgrad = devias_gemm(trans_a 0, trans_b 1, act DGELU / DRELU + aux_in, colsum = the bias gradient)
 *   devias_gemm_bias_act_wgrad = devias_gemm(trans_a 1, trans_b 1, c_f32 1, split_k, beta = gradient accumulation)
 *   devias_patch_embed_fwd     = devias_patch_im2col + devias_gemm(bias, res = positional table with res_mod);  _wgrad = the wgrad form over the same im2col matrix
 * Replaces: F.linear / nn.Linear forward, its dgrad and wgrad ATen kernels (mm / addmm) at
 *   modeling_slot.py:60-67 (Mlp), :97-101 (qkv), :113 (proj), :167-177 (Conv3d patch embed as GEMM), :302/:393 (head),
 *   :199-204 (MaskPredictor), agg_block/attention.py:66-72 (FeedForward), :108-115,123-126,141 (to_q/to_k/to_v/to_out).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* A; const void* B; void* C;
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t trans_a, trans_b;
    int32_t dtype;            /* DEVIAS_F32 | DEVIAS_BF16 : type of A, B, res, aux and (unless c_f32) C */
    int32_t c_f32;
    const float* bias;        /* [N] fp32 or NULL */
    int32_t act;              /* DEVIAS_ACT_* */
    const void* aux_in;       /* T [M,N], row stride ld_aux (DGELU / DRELU) */
    void* aux_out;            /* T [M,N], row stride ld_aux (GELU pre-activation) or NULL */
    int32_t ld_aux;
    const void* res;          /* T [*,N] residual, row stride ldr, or NULL */
    int32_t ldr, res_mod;     /* res_mod > 0: residual row = m % res_mod (positional table broadcast) */
    float beta;               /* C = acc + beta*C; only honoured for fp32 C */
    int32_t split_k;          /* >= 1 */
    float* ws;                /* split-K workspace (or colsum partials: max(M/128 * N, colsum_workspace) floats) or NULL */
    float* colsum;            /* optional fp32 [N]: colsum = colsum_beta*colsum + sum_m C[m, :] -- the bias gradient of the layer
                                 whose output gradient this GEMM produces, folded into the epilogue (split_k must be 1, C of type T) */
    float colsum_beta;
    const float* row_scale;   /* optional fp32 [ceil(M / rows_per_scale)]: (acc + bias, act) *= row_scale[m / rows_per_scale] BEFORE the residual
                                 add -- timm drop_path (stochastic depth, modeling_slot.py:36-47,150-151): 0 or 1/keep per sample */
    int32_t rows_per_scale;
    int32_t batch;            /* > 1: `batch` independent problems of the same shape in one launch; problem i uses A + i*stride_a, B + i*stride_b,
                                 C + i*stride_c (element strides).  bias / activation epilogues only (no split_k, res, aux, colsum).  Used for the
                                 per-head composite slot-attention weights and the per-clip context gradient of the folded slot attention. */
    int64_t stride_a, stride_b, stride_c;
    void* sk_ws;              /* ignored since ABI 160 (was: scratch of the stream-K schedule, retired in round 4 -- it gained nothing on any BASELINE */
    int64_t sk_ws_bytes;      /* configuration once the persistent kernel split its tail tiles); kept so that the struct layout of ABI 140-150 callers stands */
} devias_gemm_args;
int devias_gemm(const devias_gemm_args* args, void* stream);
/* bytes of workspace devias_gemm needs for the given split_k (0 when split_k <= 1) */
int64_t devias_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);

/* ---------------------------------------------------------------------------------------------------
 * Element-wise / data-movement helpers
 * ------------------------------------------------------------------------------------------------- */
/* dst[i] = (T_dst) src[i]; dtype codes as above. Used for the per-step bf16 weight copies and video input. */
int devias_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream);
/* dst[i] = (T_dst)(scale * (float) src[i]); fp32 -> fp32 (dst may be src), bf16 -> fp32, fp32 -> bf16.  The 1/world mean of the
 * all-reduced gradient buckets -- DDP's gradient averaging, run_slot_finetuning.py:552-563 -- fused with the widening of the
 * bf16 wire format; ONE launch over all buckets (they are slices of one allocation, devias_amd/parallel.py). (ABI 160) */
int devias_cast_scale(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, float scale, void* stream);
/* Tubelet im2col for PatchEmbed (nn.Conv3d k=s=(ts,ps,ps), modeling_slot.py:167-177; layout SURVEY.md §9):
 *   out[(b*Np + (t'*g + h')*g + w'), ((c*ts + kt)*ps + kh)*ps + kw] = x[b, c, ts*t'+kt, ps*h'+kh, ps*w'+kw]
 * x is [B,C,T,H,W] of dtype x_dtype, out is [B*Np, C*ts*ps*ps] of dtype out_dtype. */
int devias_patch_im2col(const void* x, int32_t x_dtype, void* out, int32_t out_dtype,
                        int32_t B, int32_t C, int32_t T, int32_t H, int32_t W, int32_t ts, int32_t ps, void* stream);
/* out[n] = beta*out[n] + sum_m x[m, n]  (bias gradients). x is T [M,N] row stride ldx, out fp32 [N].
 * ws: >= devias_colsum_workspace_bytes(M,N) bytes. */
int devias_colsum(const void* x, int32_t dtype, int32_t M, int32_t N, int32_t ldx, float* out, float beta,
                  float* ws, void* stream);
int64_t devias_colsum_workspace_bytes(int32_t M, int32_t N);
/* out[r % mod, n] (fp32 [mod,N]) = sum over rows r of x[r, n]  -- gradient of a row-broadcast (latents repeat over batch,
 * agg_block/agg_block.py:112-114).  Small inputs only (M*N <= 2^24). */
int devias_rows_reduce_mod(const void* x, int32_t dtype, int32_t M, int32_t N, int32_t mod, float* out, void* stream);
/* out[r, :] = (T) src[r % mod, :]  (src fp32 [mod,N]) */
int devias_rows_broadcast(const float* src, int32_t mod, int32_t N, void* out, int32_t dtype, int32_t M, void* stream);
/* dx = dy * f'(.) element-wise (T, n elements): act = DEVIAS_ACT_SIGMOID / DEVIAS_ACT_RELU take y = f(x) (the saved OUTPUT),
 * DEVIAS_ACT_GELU takes the saved pre-activation x. */
int devias_act_bwd(const void* dy, const void* y_or_x, void* dx, int32_t act, int32_t dtype, int64_t n, void* stream);
/* y[m, :] = x[m, :] * scale[m / rows_per_scale]  (T [M,N]); gradient of a stochastic-depth branch */
int devias_row_scale(const void* x, const float* scale, int32_t rows_per_scale, void* y, int32_t dtype, int32_t M, int32_t N, void* stream);
/* y = a * mask (+ b when b != NULL): a, b, y of dtype T, mask fp32, n elements.  Element-wise dropout with a caller-drawn mask of 0 / (1/keep)
 * (nn.Dropout before the head: fc_dropout, modeling_slot.py:291,393) and, with b, its backward plus the gradient of the un-dropped consumers */
int devias_mul_mask(const void* a, const float* mask, const void* b, void* y, int32_t dtype, int64_t n, void* stream);
/* y = a + b (same dtype T, n elements); used for gradient fan-in of the residual stream */
int devias_add(const void* a, const void* b, void* y, int32_t dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (nn.LayerNorm: modeling_slot.py:126,131,297 eps 1e-6; agg_block/attention.py:29-30,
 * agg_block/agg_block.py:105-107 eps 1e-5).  x,y: T [M,D]; gamma,beta fp32 [D]; mean,rstd fp32 [M] (saved for backward).
 * backward: dx = LN'(dy) (+ dres if dres != NULL, fusing the residual-branch gradient add);
 *           dgamma/dbeta (fp32 [D]) = beta_acc * old + column sums; ws >= devias_layernorm_bwd_workspace_bytes.
 *           dx_colsum (optional fp32 [D]) = column sums of the stored dx: the bias gradient of the Linear layer that produced x's
 *           residual branch input (proj / fc2 bias), obtained for free from the rows this kernel already holds.
 * ------------------------------------------------------------------------------------------------- */
int devias_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int32_t M, int32_t D, float eps, int32_t dtype, void* stream);
int devias_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, float* dgamma, float* dbeta, float beta_acc, float* dx_colsum,
                         int32_t M, int32_t D, int32_t dtype, float* ws, void* stream);
int64_t devias_layernorm_bwd_workspace_bytes(int32_t M, int32_t D);

/* ---------------------------------------------------------------------------------------------------
 * Encoder multi-head self-attention core (Attention.forward, modeling_slot.py:102-112): non-causal, no mask,
 * head dim 64.  qkv: T [B,N,3,H,64] exactly as F.linear produces it (:101-102, no permute copy); o: T [B,N,H*64];
 * lse: fp32 [B,H,N] = log sum_j exp(scale * q.k_j).  The N x N score matrix is never materialised.
 * backward: dqkv T [B,N,3,H,64]; delta fp32 [B,H,N] scratch (rowsum(dO*O)).  Two kernels (dQ; dK/dV), seven matrix products per tile pair (S and dP
 *   are recomputed in both so that no gradient needs a sum across workgroups: bitwise reproducible, no atomics).  `ws` (ABI 164): devias_mhsa_bwd_workspace_bytes()
 *   bytes, 16-byte aligned -- the row statistics (lse * log2 e, delta, per 32-query slice) that the dQ kernel leaves for the one-wave-per-SIMD dK / dV kernel
 *   (csrc/attn_bwd1w.hip, bf16: one wave owns a SIMD and all 512 registers, dK / dV accumulators in AGPRs, the softmax arithmetic placed in the MFMAs' gaps);
 *   with ws = NULL, or option "attn_dkdv" = 0, the two-waves-per-SIMD dK / dV kernel of rounds 2-4 runs instead (same semantics, results equal to rounding);
 *   "attn_dkdv" = 2 runs the one-wave kernel as one persistent workgroup per CU (bitwise equal to the default, one workgroup per 256-key block).
 *   CALLERS OF ABI <= 163: `ws` was documented as unused there.  It has no size argument: a non-NULL ws is WRITTEN with devias_mhsa_bwd_workspace_bytes(B, N, H)
 *   bytes (B * H * 2 * ceil(N / 32) * 32 floats), so pass either NULL or a buffer of at least that size -- never a small dummy pointer.
 * ------------------------------------------------------------------------------------------------- */
int devias_mhsa_fwd(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                    int32_t dtype, void* stream);
int devias_mhsa_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                    int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, void* ws, void* stream);
int64_t devias_mhsa_bwd_workspace_bytes(int32_t B, int32_t N, int32_t H);
/* ABI 167: the same three calls with a `flags` word (0 = the calls without it).
 * DEVIAS_ATTN_Q_PRESCALED (bf16 only): the q third of qkv holds  q' = q * scale * log2(e)  -- rounded to bf16 ONCE, by whoever produced it (devias_encoder_block_fwd: the
 * qkv GEMM on a weight copy whose first D rows and bias entries carry the factor, devias_block_args.WqkvS) -- instead of q.  Without the flag every kernel applies that
 * factor itself on operands it holds in registers: the forward and the dQ kernel round q * c, the one-wave-per-SIMD dK / dV kernel rounds k * c, so the backward's
 * scores differ from the ones the saved lse was built from by two bf16 roundings (~2x the dK / dV error at peaked logits; the advisor's round-5 finding).  With the
 * flag all three kernels multiply the SAME bf16 operands: the backward's scores ARE the forward's.  Outputs keep their meaning: o, lse (natural log, of the scores
 * scale * q k^T), and dqkv = the gradients with respect to the UNSCALED q, k, v (dQ = scale * dS k as always; dK = ln 2 * dS^T q').  Not offered with attention dropout
 * (devias_mhsa_bwd_bias_flags refuses flags != 0 with keep < 1: no forward entry point takes both). */
#define DEVIAS_ATTN_Q_PRESCALED 1
int devias_mhsa_fwd_flags(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                          int32_t dtype, int32_t flags, void* stream);
int devias_mhsa_bwd_flags(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                          int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, void* ws, int32_t flags, void* stream);
/* The same with nn.Dropout(attn_drop) on the softmax matrix (Attention.attn_drop, modeling_slot.py:90,110), ABI 161.  The matrix never exists, so neither
 * does its mask: element (b, h, query i, key j) is KEPT when  mix(rowkey + j * 0xC2B2AE35) < floor(keep * 2^32)  with
 *   rowkey = mix(mix(seed_lo ^ ((b * H + h) * 0x9E3779B1)) + seed_hi + i * 0x85EBCA6B),   mix(x): x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16
 * (32-bit wrapping arithmetic; seed_lo / seed_hi = the halves of `seed`), and kept elements are scaled by 1 / keep.  Forward and backward of one call pair
 * must be given the same (keep, seed); lse is of the un-dropped probabilities.  keep = 1 is devias_mhsa_fwd / _bwd.  (The reference draws its mask from
 * torch's generator; this one is a function of the seed so that the backward can re-create it -- oracle/ref_cpu.py attn_drop_mask restates it in numpy.) */
int devias_mhsa_fwd_dropout(const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                            int32_t dtype, float keep, uint64_t seed, void* stream);
int devias_mhsa_bwd_dropout(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                            int32_t B, int32_t N, int32_t H, float scale, int32_t dtype, float keep, uint64_t seed, void* stream);
/* Backward + the q_bias / v_bias gradients dbq, dbv: fp32 [H*64] each = the column sums of the dQ and dV thirds of dqkv over all B*N rows (the bias enters the QKV
 * projection, modeling_slot.py:97-101), ABI 162.  bf16: the two kernels emit one partial per (batch entry, 128-row block) from their fp32 accumulators and a
 * fixed-order second stage sums them -- no pass over the stored tensor (it replaces two devias_colsum calls per block: 2 x 77 MB read at ViT-B); fp32: the plain
 * backward followed by the two column sums.  keep = 1: no attention dropout (seed ignored).  ws_q, ws_v: devias_mhsa_bwd_bias_workspace_bytes() bytes each.
 * ABI 164: where devias_mhsa_bwd_bias_dv_from_do(dtype, keep) returns 1 (bf16, no dropout, option "attn_dkdv" != 0: the one-wave-per-SIMD dK / dV kernel), the
 * v_bias gradient is the column sum of d_o (every softmax row sums to one: sum_keys dV = sum_queries dO) and dbv MAY be NULL -- the caller then takes it from the
 * producer of d_o (devias_gemm's colsum epilogue on the projection's dgrad GEMM, as devias_encoder_block_bwd does); with dbv given, one column-sum pass over d_o
 * here.  The q_bias gradient still comes from the dQ kernel's accumulators. */
int32_t devias_mhsa_bwd_bias_dv_from_do(int32_t dtype, float keep);
int devias_mhsa_bwd_bias(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                         float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, void* stream);
int devias_mhsa_bwd_bias_flags(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int32_t B, int32_t N, int32_t H,
                               float scale, int32_t dtype, float keep, uint64_t seed, float* dbq, float* dbv, float* ws_q, float* ws_v, int32_t flags, void* stream);   /* ABI 167 */
int64_t devias_mhsa_bwd_bias_workspace_bytes(int32_t B, int32_t N, int32_t H);

/* ---------------------------------------------------------------------------------------------------
 * Slot cross-attention core (agg_block/attention.py:128-140): heads h, head dim dh (4 x 512 in DEVIAS),
 * softmax over the SLOT axis (:132) then renormalisation over tokens with +1e-7 (:136).
 *   q: T [B,S,h*dh]; kv: T [B,N,2,h*dh] (to_k | to_v outputs side by side, row stride 2*h*dh)
 *   attn: fp32 [B*h,S,N] = the slot-softmax A (the tensor the reference returns as `sim_distill`)
 *   rsum: fp32 [B*h,S] = sum_j A + 1e-7;  o: T [B,S,h*dh] (heads merged 'b n (h d)')
 *   ws: fp32 workspace >= devias_slot_attn_workspace_bytes
 * backward (per layer): d_o T [B,S,h*dh], d_attn_ext fp32 [B*h,S,N] or NULL (gradient arriving on the returned attn),
 *   -> dq T [B,S,h*dh], ds fp32 [B*h,S,N] (kept for the deferred K/V gradient pass).
 * kv gradient (once per distinct K/V, i.e. once for weight-tied layers): for L stacked layers
 *   dK[j] = scale * sum_l sum_i ds_l[i,j] q_l[i],  dV[j] = sum_l sum_i (A_l[i,j]/rsum_l[i]) dO_l[i]
 *   q_stack,do_stack: T [L,B,S,h*dh]; ds_stack, attn_stack: fp32 [L,B*h,S,N]; rsum_stack fp32 [L,B*h,S]; dkv: T [B,N,2,h*dh]
 * ------------------------------------------------------------------------------------------------- */
int devias_slot_attn_fwd(const void* q, const void* kv, float* attn, float* rsum, void* o,
                         int32_t B, int32_t S, int32_t N, int32_t h, int32_t dh, float scale, int32_t dtype,
                         float* ws, void* stream);
int devias_slot_attn_bwd(const void* q, const void* kv, const float* attn, const float* rsum, const void* o,
                         const void* d_o, const float* d_attn_ext, void* dq, float* ds,
                         int32_t B, int32_t S, int32_t N, int32_t h, int32_t dh, float scale, int32_t dtype,
                         float* ws, void* stream);
int devias_slot_attn_kv_grad(const void* q_stack, const void* do_stack, const float* ds_stack, const float* attn_stack,
                             const float* rsum_stack, void* dkv, int32_t L, int32_t B, int32_t S, int32_t N,
                             int32_t h, int32_t dh, float scale, int32_t dtype, void* stream);
int64_t devias_slot_attn_workspace_bytes(int32_t B, int32_t S, int32_t N, int32_t h, int32_t dh);

/* Folded slot cross-attention (same reference lines as devias_slot_attn_*: agg_block/attention.py:120-141): the to_k / to_v projections are
 * folded into the slot side, sim = scale * (Wk_h^T q) . c_j and o = Wv_h (sum_j Abar c_j), so the kernels stream the context rows
 * c = LayerNorm_ctx(features) [B*N, D] once per layer instead of K|V [B*N, 2*h*512], and no [M,D]x[D,2*h*512] GEMM exists.
 *   qp  T [B*S, h*D]  = LN(x_s) Wqk^T,  Wqk_h = Wk_h^T Wq_h  (composite weights built by the caller with batched devias_gemm)
 *   z   T [B*S, h*D]  = sum_j Abar[b,h,s,j] c[b,j,:]   (out = z Wov^T + b_o, Wov_h = Wo_h Wv_h)
 *   attn fp32 [B*h, S, N] (softmax over the slot axis, the tensor the reference returns), rsum fp32 [B*h, S] = sum_j attn + 1e-7
 * backward: dqp T [B*S, h*D], ds fp32 [B*h, S, N]; d_attn_ext (optional) = gradient arriving on the returned attention (last layer).
 * devias_slotf_pack builds, for L stacked layers sharing the context, coef T [B][K][Np] and vec T [B][K][D] (K = 2*L*h*S, Np >= N, zero padded)
 * such that the context gradient of clip b is coef[b]^T vec[b]  ([N,K] x [K,D]: one batched devias_gemm).  S <= 4, D in {384,512,768,1024}. */
int devias_slotf_fwd(const void* qp, const void* ctx, float* attn, float* rsum, void* z, int32_t B, int32_t S, int32_t N,
                     int32_t h, int32_t D, float scale, int32_t dtype, float* ws, void* stream);
int devias_slotf_bwd(const void* ctx, const float* attn, const float* rsum, const void* z, const void* dz,
                     const float* d_attn_ext, void* dqp, float* ds, int32_t B, int32_t S, int32_t N, int32_t h, int32_t D,
                     float scale, int32_t dtype, float* ws, void* stream);
int devias_slotf_pack(const float* attn_stack, const float* rsum_stack, const float* ds_stack, const void* dz_stack,
                      const void* qp_stack, void* coef, void* vec, int32_t L, int32_t B, int32_t S, int32_t N, int32_t Np,
                      int32_t h, int32_t D, float scale, int32_t dtype, void* stream);
int64_t devias_slotf_workspace_bytes(int32_t B, int32_t S, int32_t N, int32_t h, int32_t D);

/* ---------------------------------------------------------------------------------------------------
 * Slot selection (modeling_slot.py:395-406): p = softmax(slots_head); i_act[b] = argmax_s max_{c<nb} p[b,s,c];
 * i_scn[b] = argmax_s max_{nb<=c<nb+ns} p[b,s,c]; idx int32 [B,2].
 * ------------------------------------------------------------------------------------------------- */
int devias_slot_select(const void* slots_head, int32_t dtype, int32_t B, int32_t S, int32_t C, int32_t nb,
                       int32_t* idx, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * TrainLoss 'matching' branch (utils/loss/train_loss.py:85-187), one launch, no host sync:
 * per-sample S x 2 assignment (== scipy linear_sum_assignment, :112-122), CE(action), scene term by `scene_ce`:
 * 0 = scene_criterion 'KL', KL to the padded teacher logits * w_scene ('batchmean' on a 1-D input => /C, :158-164);
 * 1 = scene_criterion 'CE', cross-entropy against the teacher's argmax class, w_scene NOT applied (:155-156);
 * mask-distill MSE on the head-mean slot attention,
 * BCE-with-logits on the (already sigmoided) mask prediction, slot cosine loss.
 *   slots_head T [B*S,C]; slots T [B*S,D]; maskp T [B*S,G]; attn fp32 [B*nh,S,N]; teacher fp32 [B,ns];
 *   target int64 [B]; fg fp32 [B,G]; fgN fp32 [B,N]
 *   out_losses fp32 [6] = {action, scene, cosine, mask_prediction, mask_distill, total}
 *   out_match int32 [B,2] = (action slot i*, scene slot j*); out_logits T [B,C] = slots_head[b, i*]
 * backward (g = upstream gradient of `total`, read from device fp32 scalar g_total):
 *   d_slots_head T [B*S,C]; d_slots T [B*S,D] (cosine term only); d_maskp T [B*S,G]; d_attn fp32 [B*nh,S,N]
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, S, C, nb, ns, D, G, N, nh;
    float w_scene, w_mask_pred, w_mask_distill;
    int32_t dtype;
    int32_t scene_ce;                /* ABI 160 */
} devias_loss_dims;
int devias_head_match_loss_fwd(const devias_loss_dims* d, const void* slots_head, const void* slots, const void* maskp,
                               const float* attn, const float* teacher, const int64_t* target, const float* fg,
                               const float* fgN, float* out_losses, int32_t* out_match, void* out_logits,
                               float* ws, void* stream);
int devias_head_match_loss_bwd(const devias_loss_dims* d, const void* slots_head, const void* slots, const void* maskp,
                               const float* attn, const float* teacher, const int64_t* target, const float* fg,
                               const float* fgN, const int32_t* match, const float* g_total,
                               void* d_slots_head, void* d_slots, void* d_maskp, float* d_attn, void* stream);
int64_t devias_head_match_loss_workspace_bytes(int32_t B);

/* ---------------------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter range (torch.optim.AdamW semantics, utils/optim_factory.py:132-133;
 * decoupled weight decay, bias correction by step).  grad_scale multiplies the gradient first (1/world, loss scale).
 * ------------------------------------------------------------------------------------------------- */
int devias_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                      float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Multi-tensor optimizer step: the whole parameter list in ONE launch per operation.
 * Replaces the per-group / per-tensor Python loops of torch.optim.AdamW as created by utils/optim_factory.py:132-133 over
 * the layer-decay groups of get_parameter_groups (:49-93), the gradient norm of utils/utils.py:409-421 (get_grad_norm_:
 * 2-norm of the per-tensor 2-norms) and torch.nn.utils.clip_grad_norm_ as called at utils/utils.py:391.
 *
 * `table` is a DEVICE array of n_tensors descriptors (per tensor: pointers, length and that tensor's group
 * hyper-parameters lr = schedule·lr_scale, weight_decay, bias corrections of its step).  `chunk_tensor[c]`,
 * `chunk_index[c]` (device int32 [n_chunks]) enumerate fixed DEVIAS_OPT_CHUNK-element pieces of the tensors, one workgroup each.
 *   devias_grad_sumsq_multi: partials[c] = sum of squares of chunk c (fp32, fixed order inside the chunk)
 *   devias_clip_coef:        out[0] = sqrt(sum_c partials[c]) (fixed order), out[1] = max_norm > 0 ?
 *                            min(1, max_norm / (out[0] + 1e-6)) : 1   (clip_grad_norm_ semantics)
 *   devias_adamw_multi:      g' = g * grad_scale * (grad_scale_dev ? *grad_scale_dev : 1); AdamW update as devias_adamw_step
 * No host synchronisation anywhere: the clip coefficient stays on the device.
 * ------------------------------------------------------------------------------------------------- */
#define DEVIAS_OPT_CHUNK 16384
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    float lr, weight_decay, bc1, bc2_sqrt;   /* bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step) */
    int64_t reserved;                         /* pads the descriptor to 64 bytes */
} devias_opt_tensor;
int devias_grad_sumsq_multi(const devias_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                            int32_t n_chunks, float* partials, void* stream);
int devias_clip_coef(const float* partials, int32_t n_chunks, float max_norm, float* out, void* stream);
int devias_adamw_multi(const devias_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                       int32_t n_chunks, float beta1, float beta2, float eps, float grad_scale,
                       const float* grad_scale_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * FAME foreground masks and clip mixing (utils/transform/fame.py of the reference; CPU tensors + kornia there, called from
 * engine/engine_for_slot.py:106-108).  All buffers are device memory; images are fp32 [n, H, W].
 *   devias_fame_diff_color   video [B,3,T,H,W] (ImageNet-normalised; denormalised inside, fame.py:118) ->
 *                            diffs [B, 1 + T/2, H, W]: slot 0 = mean over the T-1 consecutive frame differences of sum_c |.|
 *                            (fame.py:93), slot 1+i = sum_c |frame 2i - frame 2i+1| (fame.py:107);
 *                            cmap  [B, H*W] int16: HSV colour bin of the temporal mean image (fame.py:47-64)
 *   devias_fame_blur         separable Gaussian, 'reflect' border, taps exp(-x^2/(2 sigma^2)) normalised (kornia GaussianBlur2d,
 *                            fame.py:20-22); in != out
 *   devias_fame_seg_refine   per image: the HW/2 largest / HW/10 smallest pixels vote into 1000-bin foreground / background
 *                            colour histograms of their clip (image i belongs to clip i / imgs_per_clip), refine =
 *                            pr_fg / (pr_bg + pr_fg) (fame.py:50-76).  Min-max normalisation (fame.py:30-36) is monotonic and
 *                            only feeds top-k selections, so it is not materialised.
 *   devias_fame_binarize_pool  per image: the num_fg largest pixels -> 1 (binmask uint8 [n,H,W], may be NULL), pooled[n, (H/pool)*(W/pool)]
 *                            = pool x pool average of the binary mask (fame.py:78-87, 142-148)
 *   devias_fame_mix          out[j] = aug[j] ? (mask[src[j]] ? video[src[j]] : video[partner[j]]) : video[src[j]]   (fame.py:124-138)
 * Top-k selections take the exact k-th value (radix select on the float bits); equal values are taken in index order.
 * ------------------------------------------------------------------------------------------------- */
int devias_fame_diff_color(const float* video, int32_t B, int32_t T, int32_t H, int32_t W, float* diffs, int16_t* cmap, void* stream);
int devias_fame_blur(const float* in, float* out, int32_t n_img, int32_t H, int32_t W, int32_t ksize, float sigma, void* stream);
int devias_fame_seg_refine(const float* blurred, const int16_t* cmap, int32_t n_img, int32_t imgs_per_clip, int32_t HW, float eps,
                           float* refine, void* stream);
int devias_fame_binarize_pool(const float* blurred, int32_t n_img, int32_t H, int32_t W, int32_t num_fg, int32_t pool,
                              uint8_t* binmask, float* pooled, void* stream);
int devias_fame_mix(const float* video, const uint8_t* binmask, int32_t mask_stride, const int32_t* src, const int32_t* partner,
                    const int32_t* aug, float* out, int32_t B, int32_t CT, int32_t HW, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused regions (ABI 150): ONE call enqueues the whole kernel sequence of a block of the step -- the same launches, in the same order and
 * with the same split-K / workspace policy as the per-kernel entry points above, so the results are bitwise theirs.  The Python host reaches
 * the library ~40 times per step instead of ~750 (one ctypes hop per fused region).  All memory is caller-owned:
 *   save     what backward re-reads (opaque layout; written by *_fwd, read by *_bwd; size from *_save_bytes)
 *   scratch  backward temporaries (dead when the call's work has run; size from *_scratch_bytes)
 *   ws       fp32 kernel workspace (split-K slabs, partial sums; size from *_workspace_bytes)
 * Gradient destinations are fp32 and may point anywhere (e.g. into a flat data-parallel gradient bucket).
 *
 * Encoder block = Block.forward of model/modeling_slot.py:142-152 (LayerNorm -> QKV Linear with q_bias | 0 | v_bias (:97-101) -> MHSA core
 * (:102-112) -> proj + residual (:113,150) -> LayerNorm -> fc1 + GELU -> fc2 + residual (:60-67,151)); stochastic depth (:36-47) through ds1 / ds2.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, N, D, H, hidden;          /* M = B*N rows; head dim D / H must be 64 */
    int32_t dtype;                       /* T: DEVIAS_F32 | DEVIAS_BF16 */
    float eps;
    const float *n1w, *n1b, *n2w, *n2b;  /* LayerNorm gamma / beta, fp32 [D] */
    const void *Wqkv, *Wp, *W1, *W2;     /* T, nn.Linear layout: [3D,D], [D,D], [hidden,D], [D,hidden] */
    const float* qkv_bias;               /* fp32 [3D] = q_bias | zeros | v_bias */
    const float *pb, *b1, *b2;           /* fp32 [D], [hidden], [D] */
    const float *ds1, *ds2;              /* optional fp32 [B]: per-sample stochastic-depth factors (0 or 1/keep) of the two branches */
    void* save;                          /* devias_encoder_block_save_bytes() */
    float* ws; int64_t ws_bytes;         /* devias_encoder_block_workspace_bytes() */
    void* sk_ws; int64_t sk_ws_bytes;    /* ignored since ABI 160 (devias_gemm_args.sk_ws) */
    /* ABI 166, optional (NULL: not used), backward only: TRANSPOSED copies of the four weights in T -- WqkvT [D,3D], WpT [D,D], W1T [D,hidden], W2T [hidden,D] -- that the
     * caller keeps beside the nn.Linear-layout copies (made once per weight update, like those).  With them the four dgrad GEMMs of a block (dY W: the reduction runs over
     * the weight's ROWS) read both operands k-contiguous instead of reading W through transposing LDS loads: the same products in the same order -- bitwise the same dx --
     * with 9-17 % less K-loop time (profiles/r6_nt_vs_tb.txt).  A caller compiled against ABI <= 165 passes a shorter struct: it must zero-extend it or keep calling an
     * ABI <= 165 library (devias_version() tells). */
    const void *WqkvT, *WpT, *W1T, *W2T;
    /* ABI 167, optional (both or neither; bf16 only): a copy of Wqkv / qkv_bias whose q third -- rows [0, D) of the matrix, entries [0, D) of the bias -- is multiplied by
     * scale * log2(e) = 0.125 * 1.4426950408889634 (in fp32, BEFORE the rounding to T).  Forward then runs the qkv GEMM on these, so that the q it leaves in the arena is
     * q' = q * scale * log2(e) rounded once, and the attention kernels of both directions run with DEVIAS_ATTN_Q_PRESCALED (see devias_mhsa_fwd_flags).  Backward reads
     * Wqkv / WqkvT (the unscaled copies) as before: dqkv holds the gradients with respect to the unscaled q, k, v.  The SAME struct must reach _fwd and _bwd of a block. */
    const void* WqkvS; const float* qkv_biasS;
} devias_block_args;
typedef struct {
    float *dn1w, *dn1b, *dWqkv, *dbqkv /* [3D]: dq_bias | (k: unused) | dv_bias */, *dWp, *dbp, *dn2w, *dn2b, *dW1, *db1, *dW2, *db2;
    float* dx_colsum;                    /* [D]: column sums of dx = the fc2-bias gradient of the PREVIOUS block (or the patch-embed bias gradient) */
    int32_t db2_done;                    /* != 0: the caller already holds colsum(dx2) (the next block's dx_colsum): db2 is not written */
    float *dbq, *dbv;                    /* ABI 160, optional: when BOTH are non-null the q_bias / v_bias gradients ([D] each: two separate parameters of the
                                            reference, modeling_slot.py:88-93) are written there -- e.g. straight into a data-parallel gradient bucket -- by two
                                            column-sum launches over the q and v thirds of dqkv (the k third has no bias), and dbqkv is not used (may be null) */
} devias_block_grads;
int64_t devias_encoder_block_save_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype);
int64_t devias_encoder_block_scratch_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype);
int64_t devias_encoder_block_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t dtype);
/* x, x2: T [B*N, D] */
int devias_encoder_block_fwd(const devias_block_args* a, const void* x, void* x2, void* stream);
/* x: the block's input again; dx2: gradient of its output; dx: gradient of its input */
int devias_encoder_block_bwd(const devias_block_args* a, const void* x, const void* dx2, void* dx, const devias_block_grads* g,
                             void* scratch, int64_t scratch_bytes, void* stream);

/* Shared head + MaskPredictor (modeling_slot.py:392-393, 194-216): Z = slots Wh^T + bh [R,C]; Mk = sigmoid(W4 relu(W2 relu(W0 slots))) [R,G].
 * R = B*S rows, h1 / h2 = the MaskPredictor's hidden widths (512, 256). */
typedef struct {
    int32_t R, D, C, h1, h2, G, dtype;
    const void *Wh, *W0, *W2, *W4;       /* T: [C,D], [h1,D], [h2,h1], [G,h2] */
    const float *bh, *b0, *b2, *b4;
    float* ws; int64_t ws_bytes;         /* devias_head_workspace_bytes() */
    const float* drop_mask;              /* optional fp32 [R, D], 0 or 1/keep: fc_dropout (nn.Dropout(fc_drop_rate), modeling_slot.py:291) applied to the
                                            slots on their way into the head ONLY (:393; the MaskPredictor and the returned features see the un-dropped slots) */
} devias_head_args;
typedef struct { float *dWh, *dbh, *dW0, *db0, *dW2, *db2, *dW4, *db4; } devias_head_grads;
int64_t devias_head_workspace_bytes(int32_t R, int32_t D, int32_t C, int32_t h1, int32_t h2, int32_t G, int32_t dtype);
int64_t devias_head_save_bytes(int32_t R, int32_t D, int32_t h1, int32_t h2, int32_t dtype);
int devias_head_fwd(const devias_head_args* a, const void* slots, void* Z, void* Mk, void* save, void* stream);
int devias_head_bwd(const devias_head_args* a, const void* slots, const void* Mk, const void* save, const void* dZ, const void* dM, void* dslots,
                    const devias_head_grads* g, void* stream);

/* Final LayerNorm + AggregationBlock with the folded slot attention (modeling_slot.py:373,381; agg_block/agg_block.py:105-139;
 * agg_block/attention.py:29-40,66-72,108-141): `depth` layers over `tied ? 1 : depth` weight sets; S <= 4 slots. */
#define DEVIAS_AGG_MAX_DEPTH 16
typedef struct {
    const void *Wq, *Wk, *Wv, *Wo, *W1, *W2;                      /* T: to_q/to_k/to_v [heads*dh, D], to_out [D, heads*dh], ff [ff,D], [D,ff] */
    const float *bo, *norm_w, *norm_b, *ctx_w, *ctx_b, *b1, *b2, *ffn_w, *ffn_b;
} devias_agg_layer_params;
typedef struct { float *dWq, *dWk, *dWv, *dWo, *dbo, *dnorm_w, *dnorm_b, *dctx_w, *dctx_b, *dW1, *db1, *dW2, *db2, *dffn_w, *dffn_b; } devias_agg_layer_grads;
typedef struct {
    int32_t B, N, S, D, depth, tied, heads, dh, ff, dtype;
    float eps_enc, eps_agg;
    const float *norm_w, *norm_b;        /* the encoder's final LayerNorm */
    const float* latents;                /* fp32 [S, D] */
    const float *last_w, *last_b;        /* last_layer LayerNorm */
    devias_agg_layer_params sets[DEVIAS_AGG_MAX_DEPTH];
    void* save;                          /* devias_agg_block_save_bytes() */
    float* ws; int64_t ws_bytes;         /* devias_agg_block_workspace_bytes() */
} devias_agg_args;
typedef struct {
    float *dnorm_w, *dnorm_b, *dlatents, *dlast_w, *dlast_b;
    float* dx_colsum;                    /* [D]: column sums of dx (the last encoder block's fc2-bias gradient) */
    devias_agg_layer_grads sets[DEVIAS_AGG_MAX_DEPTH];
} devias_agg_grads;
int64_t devias_agg_block_save_bytes(const devias_agg_args* a);
int64_t devias_agg_block_scratch_bytes(const devias_agg_args* a);
int64_t devias_agg_block_workspace_bytes(const devias_agg_args* a);
/* x: T [B*N, D] encoder output (before the final LayerNorm); slots: T [B*S, D]; *attn_out = the last layer's slot softmax, fp32 [B*heads, S, N], inside `save` */
int devias_agg_block_fwd(const devias_agg_args* a, const void* x, void* slots, float** attn_out, void* stream);
/* dslots: gradient of `slots`; dattn: optional gradient arriving on *attn_out; dx: gradient of x.  Gradients of a weight set used by several layers (tied) are the
 * sum over those layers: LayerNorm parameters accumulate layer by layer, the matrices and biases are reduced over all layers' rows in one product each after the last
 * layer has run (fixed order: deterministic) */
int devias_agg_block_bwd(const devias_agg_args* a, const void* x, const void* dslots, const float* dattn, void* dx, const devias_agg_grads* g,
                         void* scratch, int64_t scratch_bytes, void* stream);
/* the automatic split-K choices the regions (and the Python host) make, for hosts that size workspaces themselves */
int32_t devias_policy_gemm_cus(void);      /* CUs the big-tile GEMM grids and the weight-gradient split-K sizing count on: device CUs - gemm_reserve_cus, multiple of 8 */
int32_t devias_policy_small_m_split(int32_t M, int32_t N, int32_t K, int32_t trans_a);
int32_t devias_policy_wgrad_split(int32_t Nout, int32_t Kin, int32_t Mrows, int32_t dtype);

#ifdef __cplusplus
}
#endif
#endif /* DEVIAS_AMD_H */
