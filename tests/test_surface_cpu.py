"""CPU-side checks: drop-in surface, parameter-name contract, C-ABI completeness, loud failure without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from oracle import ref_cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """every function declared in include/devias_amd.h is exported by libdevias_amd.so and bound in _lib.PROTOTYPES"""
    from devias_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "devias_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int32_t|int64_t|void|const char\*)\s+(devias_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.devias_version() >= 100
    # error path: bad arguments return a code and a message, never crash (no GPU needed: validation precedes any launch)
    a = _lib.GemmArgs()
    assert lib.devias_gemm(ctypes.byref(a), None) == -1
    assert b"devias_gemm" in lib.devias_last_error()
    # launch counters and process-wide options need no GPU either
    lib.devias_counters_reset()
    assert all(lib.devias_counter(i) == 0 for i in _lib.COUNTERS.values()) and lib.devias_counter(999) == -1
    assert lib.devias_set_option(b"gemm_persistent", 1) == 0 and lib.devias_set_option(b"attn_xcd", 1) == 0
    assert lib.devias_set_option(b"no_such_option", 1) == -1 and b"no_such_option" in lib.devias_last_error()
    # the C-ABI collective validates its arguments before touching RCCL
    assert lib.devias_allreduce_bucket(None, None, 0, 0, None) == -1 and b"devias_allreduce_bucket" in lib.devias_last_error()
    lib.devias_shutdown()


def test_weight_cache_invalidation_logic(monkeypatch):
    """ADVICE r1 (high / low): compute-dtype weight copies are refreshed when Tensor._version moves (torch optimizers, load_state_dict),
    when invalidate_weight_cache() is called (the fused optimizer and `.data` writers call it: raw-pointer updates do not move the
    version), and entries die with their parameter (no id()-keyed leak)."""
    import gc
    from devias_amd import modeling_slot as ms, ops
    monkeypatch.setattr(ops, "cast", lambda t, dt, out=None: t.to(dt))            # the GPU cast kernel, stood in for on the CPU
    cache = ms._WeightCache()
    p = torch.nn.Parameter(torch.randn(8, 4))
    w0 = cache.get(p, torch.bfloat16)
    assert cache.get(p, torch.bfloat16) is w0 and cache.casts == 1
    with torch.no_grad():
        p.add_(1.0)                                                                # version bump
    w1 = cache.get(p, torch.bfloat16)
    assert w1 is not w0 and torch.equal(w1, p.detach().bfloat16()) and cache.casts == 2
    p.data.mul_(2.0)                                                               # invisible to _version ...
    assert cache.get(p, torch.bfloat16) is w1
    ms.invalidate_weight_cache()                                                   # ... which is why such writers must say so
    w2 = cache.get(p, torch.bfloat16)
    assert w2 is not w1 and torch.equal(w2, p.detach().bfloat16())
    q = torch.nn.Parameter(torch.randn(8, 4))
    c0 = cache.get_cat((p, q), torch.bfloat16)
    assert cache.get_cat((p, q), torch.bfloat16) is c0 and c0.shape == (16, 4)
    with torch.no_grad():
        q.zero_()
    assert float(cache.get_cat((p, q), torch.bfloat16)[8:].abs().max()) == 0.0
    # the transposed copies of the encoder weights (round 6: the dgrad GEMMs' k-contiguous B operand) follow the same stamps
    t0 = cache.get_t(p, torch.bfloat16)
    assert t0.shape == (4, 8) and t0.is_contiguous() and torch.equal(t0, p.detach().bfloat16().t()) and cache.get_t(p, torch.bfloat16) is t0 and cache.transposes == 1
    with torch.no_grad():
        p.add_(1.0)
    t1 = cache.get_t(p, torch.bfloat16)
    assert t1 is not t0 and torch.equal(t1, p.detach().bfloat16().t()) and cache.transposes == 2
    p.data.mul_(0.5)
    assert cache.get_t(p, torch.bfloat16) is t1
    ms.invalidate_weight_cache()
    assert torch.equal(cache.get_t(p, torch.bfloat16), p.detach().bfloat16().t()) and cache.transposes == 3
    # the q-scaled copy of a qkv weight / bias (ABI 167: DEVIAS_ATTN_Q_PRESCALED): the first `rows` rows carry the factor, applied in fp32 before the rounding; same stamps
    c_ = 0.125 * 1.4426950408889634
    s0 = cache.get_qscaled(p, 3, c_, torch.bfloat16)
    want = p.detach().clone(); want[:3] *= c_
    assert torch.equal(s0, want.bfloat16()) and cache.get_qscaled(p, 3, c_, torch.bfloat16) is s0 and cache.qscaled == 1
    assert torch.equal(s0[3:], p.detach().bfloat16()[3:]) and not torch.equal(s0[:3], p.detach().bfloat16()[:3])
    with torch.no_grad():
        p.add_(1.0)
    s1 = cache.get_qscaled(p, 3, c_, torch.bfloat16)
    want = p.detach().clone(); want[:3] *= c_
    assert s1 is not s0 and torch.equal(s1, want.bfloat16()) and cache.qscaled == 2
    bvec = torch.randn(12)                                                          # the fp32 q_bias | 0 | v_bias vector: a plain tensor, fp32 in and out
    sb = cache.get_qscaled(bvec, 4, c_, torch.float32)
    assert sb.dtype == torch.float32 and torch.equal(sb[4:], bvec[4:]) and torch.allclose(sb[:4], bvec[:4] * c_) and cache.get_qscaled(bvec, 4, c_, torch.float32) is sb
    del s0, s1, sb, bvec, want
    n = len(cache._c)
    nt_ = len(cache._t)
    del p, q, w0, w1, w2, c0, t0, t1
    gc.collect()
    assert len(cache._c) == n - 2 and len(cache._cat) == 0 and len(cache._t) == nt_ - 1
    # conv weights are viewed as matrices; fp32 mode never copies
    c = torch.nn.Parameter(torch.randn(6, 3, 2, 4, 4))
    assert cache.get(c, torch.float32).shape == (6, 96) and cache.get(c, torch.float32).data_ptr() == c.data_ptr()


@pytest.mark.parametrize("kw", [dict(), dict(num_latents=4, agg_weights_tie=False, agg_depth=4), dict(embed_dim=384, num_heads=6)])
def test_parameter_name_contract(kw):
    """named_parameters() == the reference's: checkpoint compatibility and LR-group parsing depend on it"""
    from devias_amd.modeling_slot import VisionTransformer
    cfg = ref_cpu.SlotViTConfig(all_frames=8, **kw)
    m = VisionTransformer(embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=12, qkv_bias=True, num_classes=400, all_frames=8,
                          num_latents=cfg.num_latents, agg_weights_tie=cfg.agg_weights_tie, agg_depth=cfg.agg_depth,
                          slot_matching_method="matching")
    shapes = ref_cpu.param_shapes(cfg)
    got = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert list(got) == list(shapes) and got == shapes
    sd = m.state_dict()
    assert "pos_embed" not in sd                                   # plain attribute in the reference too
    if cfg.agg_weights_tie:                                        # tied: state_dict still lists every layer (291 keys at ViT-B)
        assert f"agg_block.layers.{cfg.agg_depth - 1}.0.fn.to_q.weight" in sd
        assert sd["agg_block.layers.0.0.fn.to_q.weight"].data_ptr() == sd[f"agg_block.layers.{cfg.agg_depth - 1}.0.fn.to_q.weight"].data_ptr()
    # optim_factory.get_num_layer_for_vit parses 'blocks.<int>.' ; get_parameter_groups tests "'agg_block' in name"
    assert any(n.startswith("blocks.11.") for n in got) and any("agg_block" in n for n in got)


def test_vit_b_counts():
    from devias_amd import create_model
    m = create_model("slot_vit_base_patch16_224", num_classes=400, all_frames=16, num_latents=2, slot_matching="matching",
                     agg_weights_tie=True, agg_depth=8, drop_block_rate=None)
    assert len(m.state_dict()) == 291 and len(list(m.named_parameters())) == 186
    assert sum(p.numel() for p in m.parameters()) == 98413249


def test_mlp_head_surface():
    """head_type='mlp' (MLPHead, model/modeling_slot.py:23-34, 307-313): same parameter names and shapes as the oracle's restatement -- which
    tests/golden/vitb_t8_mlphead.npz pins to the reference -- and the reference's initialisation (fc2 scaled by init_scale)"""
    from devias_amd import create_model
    from oracle import ref_cpu
    m = create_model("slot_vit_base_patch16_224", num_classes=400, all_frames=8, num_latents=2, slot_matching="matching", agg_weights_tie=True, agg_depth=8,
                     head_type="mlp", init_scale=0.001)
    shapes = ref_cpu.param_shapes(ref_cpu.SlotViTConfig(all_frames=8, head_type="mlp"))
    assert [n for n, _ in m.named_parameters()] == list(shapes.keys())
    assert all(tuple(p.shape) == shapes[n] for n, p in m.named_parameters())
    assert float(m.head.fc2.weight.abs().max()) < 1e-3 and float(m.head.fc1.weight.abs().max()) > 1e-3
    with pytest.raises(ValueError):
        create_model("slot_vit_base_patch16_224", head_type="conv")


def test_forward_without_gpu_fails_loudly():
    from devias_amd import create_model
    m = create_model("slot_vit_small_patch16_224", num_classes=400, all_frames=8, num_latents=2, slot_matching="matching",
                     agg_weights_tie=True, agg_depth=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 224, 224))


def test_synth_is_deterministic_and_dyadic():
    from devias_amd import synth
    v = synth.video(1, 2, 32, seed=1000)
    assert torch.equal(v, synth.video(1, 2, 32, seed=1000)) and float(v.min()) >= -2 and float(v.max()) < 2
    assert torch.equal(v * 1024, (v * 1024).round())
    m196, mN = synth.fg_masks(2, 50)
    assert torch.equal(m196 * 256, (m196 * 256).round()) and torch.equal(mN.half().float(), mN)   # fp16-exact (SURVEY §8c caveat 1)
    assert torch.equal(synth.video(2, 2, 32, seed=1000, first=3)[0], synth.video(1, 2, 32, seed=1000, first=3)[0])


def test_gemm_debug_build_compiles(tmp_path):
    """the timing-ablation stamps / sentinels of gemm.hip exist only under -DDEVIAS_GEMM_DEBUG: that build must keep
    compiling for gfx950, and the production object must not contain the s_memrealtime stamps"""
    import shutil
    import subprocess
    from devias_amd import build as b
    if not os.path.exists(b.HIPCC):
        pytest.skip("hipcc not available")
    src = os.path.join(b.CSRC, "gemm.hip")
    out = tmp_path / "gemm_debug.s"
    cmd = [b.HIPCC] + b.FLAGS + ["-DDEVIAS_GEMM_DEBUG", "-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    dbg = out.read_text()
    assert "s_memrealtime" in dbg
    # production build: every stamp sits inside an #ifdef DEVIAS_GEMM_DEBUG block of the source
    depth, bad = 0, []
    for ln, line in enumerate(open(src), 1):
        t = line.strip()
        if t.startswith("#ifdef DEVIAS_GEMM_DEBUG"):
            depth += 1
        elif t.startswith("#endif") and depth:
            depth -= 1
        elif re.search(r"\bst[0-3]\s*=\s*__builtin_amdgcn_s_memrealtime", line) and depth == 0 and not t.startswith("//"):     # (the stream-K hand-off timeout also reads the clock)
            bad.append(ln)
    assert not bad, bad


def test_published_column_sums_go_stale_safely_with_two_consumers():
    """a residual-stream gradient that carries published column sums and then receives a SECOND consumer's gradient (autograd's
    InputBuffer accumulates it in place into the tagged tensor: same object, same data_ptr) must not hand out the stale sums."""
    from devias_amd import modeling_slot as ms
    seen = {}

    class Producer(torch.autograd.Function):          # stands for a block whose LayerNorm backward publishes colsum(dx) on dx
        @staticmethod
        def forward(ctx, h, tag):
            ctx.tag = tag
            return h * 2.0

        @staticmethod
        def backward(ctx, g):
            dx = g * 2.0
            ms._publish_colsum(dx, dx.sum(0))
            return dx, None

    class Upstream(torch.autograd.Function):          # stands for the previous block: consumes the (possibly accumulated) gradient
        @staticmethod
        def forward(ctx, h):
            return h + 0.0

        @staticmethod
        def backward(ctx, g):
            cs = ms._peek_colsum(g)
            seen["hit"] = cs is not None
            seen["ok"] = cs is None or torch.equal(cs, g.sum(0))
            seen["true_sum"] = g.sum(0).clone()
            return g

    h = torch.arange(12.0).reshape(4, 3).requires_grad_(True)
    y = Upstream.apply(h)
    (Producer.apply(y, 0).sum() + Producer.apply(y, 1).pow(2).sum()).backward()          # two consumers of y
    assert seen["ok"], "stale column sums were handed out"
    assert torch.equal(h.grad.sum(0), seen["true_sum"])
    # single consumer: the published sums are used
    h2 = torch.arange(12.0).reshape(4, 3).requires_grad_(True)
    Producer.apply(Upstream.apply(h2), 0).sum().backward()
    assert seen["hit"] and seen["ok"]
    # and an explicit in-place change after publication invalidates the tag
    t = torch.ones(4, 3)
    ms._publish_colsum(t, t.sum(0))
    t.add_(1.0)
    assert ms._peek_colsum(t) is None
