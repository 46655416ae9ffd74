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
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(devias_\w+)\s*\(", hdr, flags=re.M))
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
