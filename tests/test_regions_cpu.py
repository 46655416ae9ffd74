"""Host-side checks of the fused-region ABI that need no GPU: the ctypes mirrors of the argument structs have the layout the C header
declares (a C program compiled with gcc prints sizeof / offsetof), and the split-K policies the regions restate in C agree with the
Python host's own (devias_amd/ops.py) over a sweep of shapes -- the two code paths must pick the same kernels to stay bitwise equal."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STRUCTS = {"devias_block_args": "BlockArgs", "devias_block_grads": "BlockGrads", "devias_head_args": "HeadArgs", "devias_head_grads": "HeadGrads",
           "devias_agg_layer_params": "AggLayerParams", "devias_agg_layer_grads": "AggLayerGrads", "devias_agg_args": "AggArgs", "devias_agg_grads": "AggGrads",
           "devias_gemm_args": "GemmArgs", "devias_loss_dims": "LossDims"}


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    from devias_amd import _lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "devias_amd.h"', 'int main(void) {']
    for cname, pyname in STRUCTS.items():
        cls = getattr(_lib, pyname)
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for f in cls._fields_:
            lines.append(f'printf("{cname} {f[0]} %zu\\n", offsetof({cname}, {f[0]}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, pyname in STRUCTS.items():
        cls = getattr(_lib, pyname)
        assert got[(cname, "size")] == ctypes.sizeof(cls), (cname, got[(cname, "size")], ctypes.sizeof(cls))
        for f in cls._fields_:
            assert got[(cname, f[0])] == getattr(cls, f[0]).offset, (cname, f[0])


def test_split_policies_agree_with_the_python_host():
    from devias_amd import _lib, ops
    lib = _lib.load()
    dims = [1, 7, 64, 128, 196, 256, 257, 384, 512, 765, 768, 1024, 1536, 2304, 3072, 4096, 12544, 50176]
    for M in dims:
        for N in dims:
            for K in (64, 384, 512, 768, 3072, 50176):
                for ta in (0, 1):
                    tiles = ((M + 127) // 128) * ((N + 127) // 128)
                    want = max(1, min(K // 128, (256 + tiles - 1) // tiles)) if (M <= 256 and K >= 512 and not ta) else 1
                    assert lib.devias_policy_small_m_split(M, N, K, ta) == want, (M, N, K, ta)
    for Nout in dims:
        for Kin in dims:
            for Mrows in (64, 392, 512, 1568, 12544, 50176):
                for dt, bk in ((_lib.BF16, 64), (_lib.F32, 16)):
                    assert lib.devias_policy_wgrad_split(Nout, Kin, Mrows, dt) == ops.auto_split_k(Nout, Kin, Mrows, bk=bk), (Nout, Kin, Mrows, dt)


def test_region_size_queries_are_consistent():
    from devias_amd import _lib
    lib = _lib.load()
    for dt in (_lib.F32, _lib.BF16):
        assert lib.devias_encoder_block_save_bytes(2, 784, 384, 6, 1536, dt) > 0
        assert lib.devias_encoder_block_workspace_bytes(2, 784, 384, 6, 1536, dt) >= lib.devias_layernorm_bwd_workspace_bytes(2 * 784, 384)
        big, small = lib.devias_encoder_block_scratch_bytes(32, 1568, 768, 12, 3072, dt), lib.devias_encoder_block_scratch_bytes(2, 1568, 768, 12, 3072, dt)
        assert big > small > 0
    a = _lib.AggArgs()
    a.B, a.N, a.S, a.D, a.depth, a.tied, a.heads, a.dh, a.ff, a.dtype = 2, 784, 2, 384, 4, 1, 4, 512, 1536, _lib.BF16
    tied = lib.devias_agg_block_save_bytes(ctypes.byref(a))
    a.tied = 0
    assert lib.devias_agg_block_save_bytes(ctypes.byref(a)) > tied > 0          # one context + composite pair per weight set
    assert lib.devias_agg_block_scratch_bytes(ctypes.byref(a)) > 0 and lib.devias_agg_block_workspace_bytes(ctypes.byref(a)) > 0
