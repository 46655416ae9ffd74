"""Tensor-level wrappers over the C ABI (include/devias_amd.h).  PyTorch supplies device memory and the current
HIP stream; every bit of arithmetic happens in libdevias_amd.so.  No function here has a CPU or eager fallback."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F32  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_workspaces = {}


def dt_code(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"devias_amd kernels support float32 and bfloat16 tensors, got {t}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: devias_amd kernels need a GPU tensor (HIP extension only; no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous, got strides {t.stride()}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only fp32 scratch buffer per (device, stream): kernels of one stream are ordered, so a buffer is never shared by
    two launches in flight (the weight-gradient side stream gets its own)."""
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream)
    n = (int(nbytes) + 3) // 4
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws




def get_option(name: str) -> int:
    """devias_get_option: the current value of a process-wide option (to restore it after a temporary change)"""
    v = ctypes.c_int32(0)
    _lib.check(_lib.load().devias_get_option(name.encode(), ctypes.byref(v)), "devias_get_option")
    return int(v.value)


def set_option(name: str, value: int) -> None:
    """devias_set_option: process-wide kernel-selection knobs (gemm_epi, gemm256, gemm_ss, gemm_groupm, gemm_persistent, attn_cfg, attn_xcd ...)"""
    _lib.check(_lib.load().devias_set_option(name.encode(), int(value)), "devias_set_option")


def release_gemm_queue_stream(stream=None) -> None:
    """devias_gemm_release_queue_stream: the dynamic tile queues of the persistent GEMM (option gemm_dynamic) live in one ring per device that belongs to the FIRST
    stream that launched a dynamic-queue GEMM; launches on any other stream walk static tile lists (same bits, slower beside concurrent kernels).  A host that moves
    its step to another stream calls this -- after every dynamic-queue launch of the old stream has completed -- to hand the ring to `stream` (default: the current one)."""
    st = torch.cuda.current_stream() if stream is None else stream
    _lib.check(_lib.load().devias_gemm_release_queue_stream(st.cuda_stream), "devias_gemm_release_queue_stream")


def counters(reset: bool = False) -> dict:
    """launch counts per kernel family since the last reset (devias_counter): tests assert WHICH kernels served a step"""
    lib = _lib.load()
    out = {k: int(lib.devias_counter(i)) for k, i in _lib.COUNTERS.items()}
    if reset:
        lib.devias_counters_reset()
    return out


def auto_split_k(M: int, N: int, K: int, bk: int = 64) -> int:
    """Split the long reduction of a weight-gradient GEMM so that (tiles x splits) is just under ONE full round of the kernel that will run it:
    2 slots per CU for the bf16 256x256 kernel counted in 256x128 halves (512 on MI355X), 3 per CU for the 128x128 kernel (768); the CU count
    is the library's (device CUs minus the option gemm_reserve_cus), the same number the fused regions use (csrc/regions.hip)."""
    cus = _lib.load().devias_policy_gemm_cus()
    if bk == 64 and M % 256 == 0 and N % 128 == 0:
        tiles, slots = (M // 256) * (N // 128), 2 * cus
    else:
        tiles, slots = ((M + 127) // 128) * ((N + 127) // 128), 3 * cus
    if tiles >= slots or K < 8 * bk:
        return 1
    return max(1, min(slots // tiles, K // (4 * bk), 64))


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, aux_in: Optional[torch.Tensor] = None,
         aux_out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, res_mod: int = 0,
         out: Optional[torch.Tensor] = None, out_f32: bool = False, beta: float = 0.0, split_k: int = 1,
         colsum: Optional[torch.Tensor] = None, colsum_beta: float = 0.0,
         row_scale: Optional[torch.Tensor] = None, rows_per_scale: int = 0) -> torch.Tensor:
    """C[M,N] = epilogue(op(A) @ op(B)); see devias_gemm in include/devias_amd.h.
    A: [M,K] (or [K,M] if trans_a); B: [N,K] nn.Linear layout (or [K,N] if trans_b)."""
    _chk(A, "gemm.A"); _chk(B, "gemm.B", A.dtype)
    assert A.dim() == 2 and B.dim() == 2
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: reduction mismatch {K} vs {Kb}")
    cdt = torch.float32 if out_f32 else A.dtype
    if out is None:
        if beta != 0.0:
            raise ValueError("gemm: beta needs an existing `out`")
        out = torch.empty((M, N), dtype=cdt, device=A.device)
    else:
        _chk(out, "gemm.out", cdt)
        assert out.shape == (M, N)
    a = _lib.GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldc = A.shape[1], B.shape[1], N
    a.trans_a, a.trans_b = int(trans_a), int(trans_b)
    a.dtype = dt_code(A.dtype)
    a.c_f32 = int(cdt == torch.float32)
    a.bias = _p(_chk(bias, "gemm.bias", torch.float32)) if bias is not None else None
    a.act = act
    a.ld_aux = N
    if aux_in is not None:
        _chk(aux_in, "gemm.aux_in", A.dtype); assert aux_in.shape == (M, N)
        a.aux_in = aux_in.data_ptr()
    if aux_out is not None:
        _chk(aux_out, "gemm.aux_out", A.dtype); assert aux_out.shape == (M, N)
        a.aux_out = aux_out.data_ptr()
    if res is not None:
        _chk(res, "gemm.res", A.dtype)
        assert res.shape[-1] == N and res.numel() // N == (res_mod if res_mod > 0 else M)
        a.res, a.ldr, a.res_mod = res.data_ptr(), N, res_mod
    a.beta = beta
    if row_scale is not None:
        _chk(row_scale, "gemm.row_scale", torch.float32); assert rows_per_scale > 0 and row_scale.numel() * rows_per_scale >= M
        a.row_scale, a.rows_per_scale = row_scale.data_ptr(), rows_per_scale
    if split_k == 1 and M <= 256 and K >= 512 and not trans_a:
        # small-M (B*S-row slot MLP) GEMMs are latency bound on a handful of tiles: split K to fill the chip
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        split_k = max(1, min(K // 128, (256 + tiles - 1) // tiles))
    if colsum is not None:            # bias gradient folded into the epilogue (needs the un-split kernel)
        _chk(colsum, "gemm.colsum", torch.float32); assert colsum.numel() == N
        split_k = 1
        a.colsum, a.colsum_beta = colsum.data_ptr(), colsum_beta
        nbytes = max(((M + 127) // 128) * N * 4, _lib.load().devias_colsum_workspace_bytes(M, N))
        a.ws = workspace(nbytes, A.device).data_ptr()
    a.split_k = split_k
    if split_k > 1:
        nbytes = _lib.load().devias_gemm_workspace_bytes(M, N, split_k)
        a.ws = workspace(nbytes, A.device).data_ptr()
    _lib.check(_lib.load().devias_gemm(ctypes.byref(a), _stream()), "devias_gemm")
    return out


def wgrad(dY: torch.Tensor, X: torch.Tensor, out: Optional[torch.Tensor] = None, beta: float = 0.0) -> torch.Tensor:
    """dW[N,K] (fp32) = dY[M,N]^T @ X[M,K]   (weight gradient of Y = X W^T); split-K over the long M reduction."""
    M, N = dY.shape
    K = X.shape[1]
    sk = auto_split_k(N, K, M, bk=64 if dY.dtype == torch.bfloat16 else 16)     # C is [N, K], the reduction runs over M
    return gemm(dY, X, trans_a=True, trans_b=True, out=out, out_f32=True, beta=beta, split_k=sk)


def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None, scale: Optional[float] = None) -> torch.Tensor:
    """out = (dtype) src, or (dtype)(scale * src) when `scale` is given (out may be src for fp32 -> fp32)."""
    _chk(src, "cast.src")
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    _chk(out, "cast.out", dtype)
    if scale is not None:
        _lib.check(_lib.load().devias_cast_scale(src.data_ptr(), dt_code(src.dtype), out.data_ptr(), dt_code(dtype), src.numel(), float(scale), _stream()),
                   "devias_cast_scale")
        return out
    _lib.check(_lib.load().devias_cast(src.data_ptr(), dt_code(src.dtype), out.data_ptr(), dt_code(dtype), src.numel(), _stream()),
               "devias_cast")
    return out


def patch_im2col(x: torch.Tensor, tubelet: int, patch: int, dtype: torch.dtype) -> torch.Tensor:
    _chk(x, "patch_im2col.x")
    B, C, T, H, W = x.shape
    n_tok = (T // tubelet) * (H // patch) * (W // patch)
    out = torch.empty((B * n_tok, C * tubelet * patch * patch), dtype=dtype, device=x.device)
    _lib.check(_lib.load().devias_patch_im2col(x.data_ptr(), dt_code(x.dtype), out.data_ptr(), dt_code(dtype), B, C, T, H, W,
                                               tubelet, patch, _stream()), "devias_patch_im2col")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, beta: float = 0.0, cols: Optional[tuple] = None) -> torch.Tensor:
    """out[n] = beta * out[n] + sum_m x[m, n]; `cols = (first, count)` sums only that column range of the row-major matrix (row stride = its full width)"""
    _chk(x, "colsum.x")
    M, ld = x.shape
    c0, N = cols if cols is not None else (0, ld)
    assert 0 <= c0 and c0 + N <= ld
    if out is None:
        assert beta == 0.0
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    ws = workspace(_lib.load().devias_colsum_workspace_bytes(M, N), x.device)
    _lib.check(_lib.load().devias_colsum(x.data_ptr() + c0 * x.element_size(), dt_code(x.dtype), M, N, ld, out.data_ptr(), beta, ws.data_ptr(), _stream()),
               "devias_colsum")
    return out


def rows_reduce_mod(x: torch.Tensor, mod: int) -> torch.Tensor:
    _chk(x, "rows_reduce_mod.x")
    M, N = x.shape
    out = torch.empty((mod, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().devias_rows_reduce_mod(x.data_ptr(), dt_code(x.dtype), M, N, mod, out.data_ptr(), _stream()),
               "devias_rows_reduce_mod")
    return out


def rows_broadcast(src: torch.Tensor, M: int, dtype: torch.dtype) -> torch.Tensor:
    _chk(src, "rows_broadcast.src", torch.float32)
    mod, N = src.shape
    out = torch.empty((M, N), dtype=dtype, device=src.device)
    _lib.check(_lib.load().devias_rows_broadcast(src.data_ptr(), mod, N, out.data_ptr(), dt_code(dtype), M, _stream()),
               "devias_rows_broadcast")
    return out


def act_bwd(dy: torch.Tensor, y: torch.Tensor, act: int) -> torch.Tensor:
    _chk(dy, "act_bwd.dy"); _chk(y, "act_bwd.y", dy.dtype)
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().devias_act_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), act, dt_code(dy.dtype), dy.numel(), _stream()),
               "devias_act_bwd")
    return dx


def mul_mask(a: torch.Tensor, mask: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = a * mask (+ b): element-wise dropout with a caller-drawn fp32 mask of 0 / (1/keep) (devias_mul_mask)"""
    _chk(a, "mul_mask.a"); _chk(mask, "mul_mask.mask", torch.float32)
    assert mask.numel() == a.numel()
    if b is not None:
        _chk(b, "mul_mask.b", a.dtype)
    y = torch.empty_like(a)
    _lib.check(_lib.load().devias_mul_mask(a.data_ptr(), mask.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), dt_code(a.dtype), a.numel(), _stream()),
               "devias_mul_mask")
    return y


def row_scale(x: torch.Tensor, scale: torch.Tensor, rows_per_scale: int) -> torch.Tensor:
    _chk(x, "row_scale.x"); _chk(scale, "row_scale.scale", torch.float32)
    M, N = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().devias_row_scale(x.data_ptr(), scale.data_ptr(), rows_per_scale, y.data_ptr(), dt_code(x.dtype), M, N, _stream()),
               "devias_row_scale")
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _chk(a, "add.a"); _chk(b, "add.b", a.dtype)
    y = torch.empty_like(a)
    _lib.check(_lib.load().devias_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), dt_code(a.dtype), a.numel(), _stream()), "devias_add")
    return y


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    _chk(x, "layernorm_fwd.x"); _chk(gamma, "gamma", torch.float32); _chk(beta, "beta", torch.float32)
    M, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().devias_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), M, D, eps, dt_code(x.dtype), _stream()), "devias_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None, beta_acc: float = 0.0, dx_colsum=None):
    _chk(dy, "layernorm_bwd.dy"); _chk(x, "layernorm_bwd.x", dy.dtype)
    M, D = x.shape
    dx = torch.empty_like(x)
    if dgamma is None or dbeta is None:
        assert beta_acc == 0.0
        dgamma = dgamma if dgamma is not None else torch.empty((D,), dtype=torch.float32, device=x.device)
        dbeta = dbeta if dbeta is not None else torch.empty((D,), dtype=torch.float32, device=x.device)
    _chk(dgamma, "layernorm_bwd.dgamma", torch.float32); _chk(dbeta, "layernorm_bwd.dbeta", torch.float32)
    if dres is not None:
        _chk(dres, "layernorm_bwd.dres", dy.dtype)
    ws = workspace(_lib.load().devias_layernorm_bwd_workspace_bytes(M, D), x.device)
    _lib.check(_lib.load().devias_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                _p(dres), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), beta_acc, _p(dx_colsum), M, D,
                                                dt_code(x.dtype), ws.data_ptr(), _stream()), "devias_layernorm_bwd")
    return dx, dgamma, dbeta


def mhsa_fwd(qkv: torch.Tensor, B: int, N: int, H: int, scale: float, out: Optional[torch.Tensor] = None, drop=None, q_prescaled: bool = False):
    """drop = (keep, seed): nn.Dropout(1 - keep) on the softmax matrix, mask = the library's hash of (seed, b, h, i, j) (include/devias_amd.h)"""
    _chk(qkv, "mhsa_fwd.qkv")
    assert qkv.numel() == B * N * 3 * H * 64, "mhsa: head dim must be 64"
    if out is None:
        o = torch.empty((B * N, H * 64), dtype=qkv.dtype, device=qkv.device)
    else:
        o = _chk(out, "mhsa_fwd.out", qkv.dtype)
        assert o.shape == (B * N, H * 64)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    assert not (q_prescaled and drop is not None and float(drop[0]) < 1.0), "mhsa_fwd: q_prescaled is not offered with attention dropout"
    if drop is not None and float(drop[0]) < 1.0:
        _lib.check(_lib.load().devias_mhsa_fwd_dropout(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, scale, dt_code(qkv.dtype),
                                                       float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF, _stream()), "devias_mhsa_fwd_dropout")
        return o, lse
    if q_prescaled:          # the q third of qkv holds q * scale * log2(e) (DEVIAS_ATTN_Q_PRESCALED, bf16; not with attention dropout)
        _lib.check(_lib.load().devias_mhsa_fwd_flags(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, scale, dt_code(qkv.dtype), _lib.ATTN_Q_PRESCALED, _stream()),
                   "devias_mhsa_fwd_flags")
        return o, lse
    _lib.check(_lib.load().devias_mhsa_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, scale, dt_code(qkv.dtype), _stream()),
               "devias_mhsa_fwd")
    return o, lse


def mhsa_bwd_dv_from_do(dtype: torch.dtype, drop=None) -> bool:
    """True where the v_bias gradient is the column sum of d_o (devias_mhsa_bwd_bias_dv_from_do: bf16, no attention dropout, one-wave-per-SIMD dK / dV kernel):
    the caller asks the GEMM that produces d_o for its column sums and passes bias_out = (dbq, None)"""
    keep = float(drop[0]) if drop is not None else 1.0
    return bool(_lib.load().devias_mhsa_bwd_bias_dv_from_do(dt_code(dtype), keep))


def mhsa_bwd(qkv, o, d_o, lse, B: int, N: int, H: int, scale: float, drop=None, bias_out=None, q_prescaled: bool = False):
    """bias_out = (dbq, dbv): fp32 [H * 64] destinations of the q_bias / v_bias gradients (column sums of dQ / dV over all rows), produced by the same call
    (devias_mhsa_bwd_bias: from the kernels' accumulators in bf16, by two column-sum passes in fp32)"""
    _chk(qkv, "mhsa_bwd.qkv"); _chk(o, "mhsa_bwd.o", qkv.dtype); _chk(d_o, "mhsa_bwd.d_o", qkv.dtype)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    if bias_out is not None:
        dbq, dbv = (None if t is None else _chk(t, "mhsa_bwd.bias_out", torch.float32) for t in bias_out)
        assert dbq.numel() == H * 64 and (dbv is None or dbv.numel() == H * 64)      # (dbv = None: only where mhsa_bwd_dv_from_do() says the caller has it from colsum(d_o))
        wsb = int(_lib.load().devias_mhsa_bwd_bias_workspace_bytes(B, N, H))
        ws = torch.empty((2, (wsb + 3) // 4), dtype=torch.float32, device=qkv.device)
        keep, seed = (float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF) if drop is not None else (1.0, 0)
        if q_prescaled:
            _lib.check(_lib.load().devias_mhsa_bwd_bias_flags(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, N, H, scale,
                                                              dt_code(qkv.dtype), keep, seed, dbq.data_ptr(), None if dbv is None else dbv.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                                              _lib.ATTN_Q_PRESCALED, _stream()), "devias_mhsa_bwd_bias_flags")
            return dqkv
        _lib.check(_lib.load().devias_mhsa_bwd_bias(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, N, H, scale,
                                                    dt_code(qkv.dtype), keep, seed, dbq.data_ptr(), None if dbv is None else dbv.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), _stream()),
                   "devias_mhsa_bwd_bias")
        return dqkv
    assert not (q_prescaled and drop is not None and float(drop[0]) < 1.0), "mhsa_bwd: q_prescaled is not offered with attention dropout"
    if drop is not None and float(drop[0]) < 1.0:
        _lib.check(_lib.load().devias_mhsa_bwd_dropout(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                                                       B, N, H, scale, dt_code(qkv.dtype), float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF, _stream()),
                   "devias_mhsa_bwd_dropout")
        return dqkv
    wsb = int(_lib.load().devias_mhsa_bwd_workspace_bytes(B, N, H))      # row statistics of the one-wave-per-SIMD dK / dV kernel (bf16)
    ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=qkv.device)
    if q_prescaled:
        _lib.check(_lib.load().devias_mhsa_bwd_flags(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                                     dqkv.data_ptr(), B, N, H, scale, dt_code(qkv.dtype), ws.data_ptr(), _lib.ATTN_Q_PRESCALED, _stream()), "devias_mhsa_bwd_flags")
        return dqkv
    _lib.check(_lib.load().devias_mhsa_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                           dqkv.data_ptr(), B, N, H, scale, dt_code(qkv.dtype), ws.data_ptr(), _stream()), "devias_mhsa_bwd")
    return dqkv


def slot_attn_fwd(q, kv, B, S, N, h, dh, scale, attn_out=None, rsum_out=None):
    _chk(q, "slot_attn_fwd.q"); _chk(kv, "slot_attn_fwd.kv", q.dtype)
    dev = q.device
    attn = attn_out if attn_out is not None else torch.empty((B * h, S, N), dtype=torch.float32, device=dev)
    rsum = rsum_out if rsum_out is not None else torch.empty((B * h, S), dtype=torch.float32, device=dev)
    o = torch.empty((B * S, h * dh), dtype=q.dtype, device=dev)
    ws = workspace(_lib.load().devias_slot_attn_workspace_bytes(B, S, N, h, dh), dev)
    _lib.check(_lib.load().devias_slot_attn_fwd(q.data_ptr(), kv.data_ptr(), attn.data_ptr(), rsum.data_ptr(), o.data_ptr(), B, S, N,
                                                h, dh, scale, dt_code(q.dtype), ws.data_ptr(), _stream()), "devias_slot_attn_fwd")
    return attn, rsum, o


def slot_attn_bwd(q, kv, attn, rsum, o, d_o, d_attn_ext, B, S, N, h, dh, scale, ds_out=None):
    dev = q.device
    _chk(d_o, "slot_attn_bwd.d_o", q.dtype)
    dq = torch.empty_like(q)
    ds = ds_out if ds_out is not None else torch.empty((B * h, S, N), dtype=torch.float32, device=dev)
    if d_attn_ext is not None:
        _chk(d_attn_ext, "slot_attn_bwd.d_attn_ext", torch.float32)
    ws = workspace(_lib.load().devias_slot_attn_workspace_bytes(B, S, N, h, dh), dev)
    _lib.check(_lib.load().devias_slot_attn_bwd(q.data_ptr(), kv.data_ptr(), attn.data_ptr(), rsum.data_ptr(), o.data_ptr(),
                                                d_o.data_ptr(), _p(d_attn_ext), dq.data_ptr(), ds.data_ptr(), B, S, N, h, dh, scale,
                                                dt_code(q.dtype), ws.data_ptr(), _stream()), "devias_slot_attn_bwd")
    return dq, ds


def slot_attn_kv_grad(q_stack, do_stack, ds_stack, attn_stack, rsum_stack, L, B, S, N, h, dh, scale):
    for t, n in ((q_stack, "q_stack"), (do_stack, "do_stack"), (ds_stack, "ds_stack"), (attn_stack, "attn_stack"),
                 (rsum_stack, "rsum_stack")):
        _chk(t, "slot_attn_kv_grad." + n)
    dkv = torch.empty((B * N, 2 * h * dh), dtype=q_stack.dtype, device=q_stack.device)
    _lib.check(_lib.load().devias_slot_attn_kv_grad(q_stack.data_ptr(), do_stack.data_ptr(), ds_stack.data_ptr(), attn_stack.data_ptr(),
                                                    rsum_stack.data_ptr(), dkv.data_ptr(), L, B, S, N, h, dh, scale,
                                                    dt_code(q_stack.dtype), _stream()), "devias_slot_attn_kv_grad")
    return dkv


def gemm_batched(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, *, lda: int, ldb: int, ldc: int,
                 stride_a: int, stride_b: int, stride_c: int, batch: int, trans_a: bool = False, trans_b: bool = False,
                 a_off: int = 0, b_off: int = 0, c_off: int = 0) -> torch.Tensor:
    """`batch` independent C_i[M,N] = op(A_i) @ op(B_i) in one launch (devias_gemm with batch > 1): problem i reads A + a_off + i*stride_a ...
    (element offsets / strides into the given tensors, which only supply base pointers and dtypes).  C may be fp32 with bf16 operands."""
    _chk(A, "gemm_batched.A"); _chk(B, "gemm_batched.B", A.dtype); _chk(C, "gemm_batched.C")
    if C.dtype not in (A.dtype, torch.float32):
        raise TypeError("gemm_batched: C must have the operand dtype or float32")
    a = _lib.GemmArgs()
    es, ec = A.element_size(), C.element_size()
    a.A, a.B, a.C = A.data_ptr() + a_off * es, B.data_ptr() + b_off * es, C.data_ptr() + c_off * ec
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldc = lda, ldb, ldc
    a.trans_a, a.trans_b = int(trans_a), int(trans_b)
    a.dtype = dt_code(A.dtype)
    a.c_f32 = int(C.dtype == torch.float32)
    a.split_k = 1
    a.batch, a.stride_a, a.stride_b, a.stride_c = batch, stride_a, stride_b, stride_c
    _lib.check(_lib.load().devias_gemm(ctypes.byref(a), _stream()), "devias_gemm(batched)")
    return C


def slotf_fwd(qp, ctx, B, S, N, h, D, scale, attn_out=None, rsum_out=None):
    """folded slot attention forward (devias_slotf_fwd): qp [B*S, h*D], ctx [B*N, D] -> attn fp32 [B*h,S,N], rsum fp32 [B*h,S], z [B*S, h*D]"""
    _chk(qp, "slotf_fwd.qp"); _chk(ctx, "slotf_fwd.ctx", qp.dtype)
    dev = qp.device
    attn = attn_out if attn_out is not None else torch.empty((B * h, S, N), dtype=torch.float32, device=dev)
    rsum = rsum_out if rsum_out is not None else torch.empty((B * h, S), dtype=torch.float32, device=dev)
    z = torch.empty((B * S, h * D), dtype=qp.dtype, device=dev)
    ws = workspace(_lib.load().devias_slotf_workspace_bytes(B, S, N, h, D), dev)
    _lib.check(_lib.load().devias_slotf_fwd(qp.data_ptr(), ctx.data_ptr(), attn.data_ptr(), rsum.data_ptr(), z.data_ptr(), B, S, N, h, D,
                                            scale, dt_code(qp.dtype), ws.data_ptr(), _stream()), "devias_slotf_fwd")
    return attn, rsum, z


def slotf_bwd(ctx, attn, rsum, z, dz, d_attn_ext, B, S, N, h, D, scale, ds_out=None):
    _chk(ctx, "slotf_bwd.ctx"); _chk(z, "slotf_bwd.z", ctx.dtype); _chk(dz, "slotf_bwd.dz", ctx.dtype)
    dev = ctx.device
    dqp = torch.empty((B * S, h * D), dtype=ctx.dtype, device=dev)
    ds = ds_out if ds_out is not None else torch.empty((B * h, S, N), dtype=torch.float32, device=dev)
    if d_attn_ext is not None:
        _chk(d_attn_ext, "slotf_bwd.d_attn_ext", torch.float32)
    ws = workspace(_lib.load().devias_slotf_workspace_bytes(B, S, N, h, D), dev)
    _lib.check(_lib.load().devias_slotf_bwd(ctx.data_ptr(), attn.data_ptr(), rsum.data_ptr(), z.data_ptr(), dz.data_ptr(), _p(d_attn_ext),
                                            dqp.data_ptr(), ds.data_ptr(), B, S, N, h, D, scale, dt_code(ctx.dtype), ws.data_ptr(), _stream()),
               "devias_slotf_bwd")
    return dqp, ds


def slotf_context_grad(attn_stack, rsum_stack, ds_stack, dz_stack, qp_stack, L, B, S, N, h, D, scale):
    """dc [B*N, D] of L stacked folded layers that share the context rows: devias_slotf_pack + one batched [N,K] x [K,D] GEMM per clip"""
    for t, n in ((attn_stack, "attn_stack"), (rsum_stack, "rsum_stack"), (ds_stack, "ds_stack"), (dz_stack, "dz_stack"), (qp_stack, "qp_stack")):
        _chk(t, "slotf_context_grad." + n)
    dt, dev = qp_stack.dtype, qp_stack.device
    K = 2 * L * h * S
    Np = (N + 7) // 8 * 8
    coef = torch.empty((B, K, Np), dtype=dt, device=dev)
    vec = torch.empty((B, K, D), dtype=dt, device=dev)
    _lib.check(_lib.load().devias_slotf_pack(attn_stack.data_ptr(), rsum_stack.data_ptr(), ds_stack.data_ptr(), dz_stack.data_ptr(),
                                             qp_stack.data_ptr(), coef.data_ptr(), vec.data_ptr(), L, B, S, N, Np, h, D, scale, dt_code(dt),
                                             _stream()), "devias_slotf_pack")
    dc = torch.empty((B * N, D), dtype=dt, device=dev)
    gemm_batched(coef, vec, dc, N, D, K, lda=Np, ldb=D, ldc=D, stride_a=K * Np, stride_b=K * D, stride_c=N * D, batch=B,
                 trans_a=True, trans_b=True)
    return dc


def slot_select(slots_head: torch.Tensor, B: int, S: int, nb: int) -> torch.Tensor:
    _chk(slots_head, "slot_select.slots_head")
    C = slots_head.shape[-1]
    idx = torch.empty((B, 2), dtype=torch.int32, device=slots_head.device)
    _lib.check(_lib.load().devias_slot_select(slots_head.data_ptr(), dt_code(slots_head.dtype), B, S, C, nb, idx.data_ptr(), _stream()),
               "devias_slot_select")
    return idx


def _loss_dims(slots_head, slots, maskp, attn, teacher, B, nb, w_scene, w_mp, w_md, scene_ce=False):
    d = _lib.LossDims()
    d.B = B
    d.S = slots_head.shape[0] // B
    d.C = slots_head.shape[1]
    d.nb, d.ns = nb, teacher.shape[1]
    d.D, d.G, d.N = slots.shape[1], maskp.shape[1], attn.shape[2]
    d.nh = attn.shape[0] // B
    d.w_scene, d.w_mask_pred, d.w_mask_distill = w_scene, w_mp, w_md
    d.dtype = dt_code(slots_head.dtype)
    d.scene_ce = 1 if scene_ce else 0
    return d


def head_match_loss_fwd(slots_head, slots, maskp, attn, teacher, target, fg, fgN, nb, w_scene, w_mp, w_md, scene_ce=False):
    B = target.shape[0]
    for t, n, dt in ((slots_head, "slots_head", None), (slots, "slots", slots_head.dtype), (maskp, "maskp", slots_head.dtype),
                     (attn, "attn", torch.float32), (teacher, "teacher", torch.float32), (target, "target", torch.int64),
                     (fg, "fg", torch.float32), (fgN, "fgN", torch.float32)):
        _chk(t, "head_match_loss_fwd." + n, dt)
    d = _loss_dims(slots_head, slots, maskp, attn, teacher, B, nb, w_scene, w_mp, w_md, scene_ce)
    dev = slots_head.device
    losses = torch.empty((6,), dtype=torch.float32, device=dev)
    match = torch.empty((B, 2), dtype=torch.int32, device=dev)
    logits = torch.empty((B, d.C), dtype=slots_head.dtype, device=dev)
    ws = workspace(_lib.load().devias_head_match_loss_workspace_bytes(B), dev)
    _lib.check(_lib.load().devias_head_match_loss_fwd(ctypes.byref(d), slots_head.data_ptr(), slots.data_ptr(), maskp.data_ptr(),
                                                      attn.data_ptr(), teacher.data_ptr(), target.data_ptr(), fg.data_ptr(),
                                                      fgN.data_ptr(), losses.data_ptr(), match.data_ptr(), logits.data_ptr(),
                                                      ws.data_ptr(), _stream()), "devias_head_match_loss_fwd")
    return losses, match, logits


def head_match_loss_bwd(slots_head, slots, maskp, attn, teacher, target, fg, fgN, match, g_total, nb, w_scene, w_mp, w_md, scene_ce=False):
    B = target.shape[0]
    _chk(g_total, "head_match_loss_bwd.g_total", torch.float32)
    d = _loss_dims(slots_head, slots, maskp, attn, teacher, B, nb, w_scene, w_mp, w_md, scene_ce)
    dZ = torch.empty_like(slots_head)
    dslots = torch.empty_like(slots)
    dmask = torch.empty_like(maskp)
    dattn = torch.empty_like(attn)
    _lib.check(_lib.load().devias_head_match_loss_bwd(ctypes.byref(d), slots_head.data_ptr(), slots.data_ptr(), maskp.data_ptr(),
                                                      attn.data_ptr(), teacher.data_ptr(), target.data_ptr(), fg.data_ptr(),
                                                      fgN.data_ptr(), match.data_ptr(), g_total.data_ptr(), dZ.data_ptr(),
                                                      dslots.data_ptr(), dmask.data_ptr(), dattn.data_ptr(), _stream()),
               "devias_head_match_loss_bwd")
    return dZ, dslots, dmask, dattn


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, "adamw_step." + n, torch.float32)
    _lib.check(_lib.load().devias_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(),
                                             lr, beta1, beta2, eps, weight_decay, step, grad_scale, _stream()), "devias_adamw_step")


def grad_sumsq_multi(table, chunk_tensor, chunk_index, partials):
    """devias_grad_sumsq_multi: table uint8 [n_tensors, 64] (devias_opt_tensor records), chunk lists int32, partials fp32."""
    _chk(table, "grad_sumsq_multi.table", torch.uint8); _chk(partials, "grad_sumsq_multi.partials", torch.float32)
    _chk(chunk_tensor, "grad_sumsq_multi.chunk_tensor", torch.int32); _chk(chunk_index, "grad_sumsq_multi.chunk_index", torch.int32)
    _lib.check(_lib.load().devias_grad_sumsq_multi(table.data_ptr(), chunk_tensor.data_ptr(), chunk_index.data_ptr(), chunk_tensor.numel(),
                                                   partials.data_ptr(), _stream()), "devias_grad_sumsq_multi")


def clip_coef(partials, n_chunks, max_norm, out):
    _chk(partials, "clip_coef.partials", torch.float32); _chk(out, "clip_coef.out", torch.float32)
    _lib.check(_lib.load().devias_clip_coef(partials.data_ptr(), n_chunks, float(max_norm), out.data_ptr(), _stream()), "devias_clip_coef")


def adamw_multi(table, chunk_tensor, chunk_index, beta1, beta2, eps, grad_scale=1.0, grad_scale_dev=None):
    _chk(table, "adamw_multi.table", torch.uint8)
    _chk(chunk_tensor, "adamw_multi.chunk_tensor", torch.int32); _chk(chunk_index, "adamw_multi.chunk_index", torch.int32)
    if grad_scale_dev is not None:
        _chk(grad_scale_dev, "adamw_multi.grad_scale_dev", torch.float32)
    _lib.check(_lib.load().devias_adamw_multi(table.data_ptr(), chunk_tensor.data_ptr(), chunk_index.data_ptr(), chunk_tensor.numel(),
                                              beta1, beta2, eps, grad_scale, _p(grad_scale_dev), _stream()), "devias_adamw_multi")
