"""The fused regions (devias_encoder_block_* / devias_agg_block_* / devias_head_*, one library call per region and direction) against the
per-kernel path they replace: the SAME launches in the same order, so outputs and every parameter gradient must be BITWISE equal
(one exception since the agg block's region defers the weight / bias gradients of a TIED weight set until all its layers have run and reduces over
all their rows in one product -- the per-kernel path adds one product per layer: those gradients agree to fp32 summation order) --
fp32 and bf16, weight-tied and untied aggregation block, stochastic depth (same masks), gradient accumulation into existing .grad, and
with the data-parallel gradient buckets attached (weight gradients written straight into the bucket views)."""
import pytest
import torch

import golden_util as gu
from devias_amd import synth

pytestmark = pytest.mark.gpu


def _build(cfg, dtype, **kw):
    from functools import partial
    from devias_amd.modeling_slot import VisionTransformer
    m = VisionTransformer(img_size=cfg.img_size, patch_size=16, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4,
                          qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=cfg.num_classes,
                          all_frames=cfg.all_frames, tubelet_size=cfg.tubelet_size, init_scale=1e-3,
                          num_latents=cfg.num_latents, head_type="linear", slot_matching_method="matching",
                          agg_weights_tie=cfg.agg_weights_tie, agg_depth=cfg.agg_depth,
                          num_scene_classes=cfg.num_scene_classes, compute_dtype=dtype, **kw)
    synth.fill_module_(m, seed=0)
    return m.cuda().train()


def _crit():
    from devias_amd.train_loss import TrainLoss
    return TrainLoss(criterion=None, scene_criterion="KL", num_action_classes=400, slot_matching_method="matching",
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0, scene_loss_weight=4000, sync_loss_dict=False)


def _step(model, crit, data, regions, seed=None, zero=True):
    import devias_amd.modeling_slot as ms
    x, y, tl, fg = data
    old = ms._REGIONS
    ms._REGIONS = regions
    try:
        if zero:
            for p in model.parameters():
                p.grad = None
        if seed is not None:
            torch.manual_seed(seed)
        out = model(x)
        total, logits, ld = crit(model, out, (None, tl), y, fg_mask=fg)
        total.backward()
        torch.cuda.synchronize()
    finally:
        ms._REGIONS = old
    (af, sf), (al, sl, attn), (sh, slots, mk) = out
    outs = {"total": total.detach().clone(), "slots_head": sh.detach().clone(), "slots": slots.detach().clone(), "mask": mk.detach().clone(),
            "attn": attn.detach().clone(), "action_logit": al.detach().clone()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return outs, grads


def _data(cfg, B):
    x, y, tl, fg = gu.inputs(cfg, B)
    return x.cuda(), y.cuda(), tl.cuda(), (fg[0].cuda(), fg[1].cuda())


# the agg block's matrices and biases whose gradients the region reduces over all layers of a tied weight set at once (csrc/regions.hip: devias_agg_block_bwd)
_DEFERRED = ("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias", ".net.0.weight", ".net.0.bias", ".net.3.weight", ".net.3.bias",
             ".norm.weight", ".norm.bias")       # (round 5: also the two pre-norms' parameters -- partials of all layers, one reduce per parameter)


def _assert_bitwise(a, b, what, order_tol=0.0):
    bad = []
    for k in a:
        if torch.equal(a[k], b[k]):
            continue
        if order_tol and "agg_block" in k and k.endswith(_DEFERRED):
            # relative to the tensor's own scale -- or, for the attention pre-norm's bias, to its weight gradient's: the slot softmax does not see a shift common
            # to all slot queries, so that bias gradient is rounding noise around zero in either summation order
            scale = a[k].float().abs().max().item()
            if k.endswith(".0.norm.bias"):
                scale = max(scale, a[k[:-4] + "weight"].float().abs().max().item())
            err = (a[k].float() - b[k].float()).abs().max().item() / (scale + 1e-30)
            if err <= order_tol:
                continue
            k = f"{k} (err {err:.2e}, scale {scale:.2e})"
        bad.append(k)
    assert not bad, f"{what}: not bitwise equal: {bad[:6]} ({len(bad)} of {len(a)})"


@pytest.mark.parametrize("name,dtype,B", [("vits_t8", "fp32", 2), ("vits_t8", "bf16", 2), ("vitb_t8_s4_untied", "fp32", 2), ("vitb_t8_s4_untied", "bf16", 2),
                                          ("vitb_t16", "bf16", 8)])
def test_regions_bitwise_equal_per_kernel_path(name, dtype, B):
    from devias_amd import ops
    fx, cfg, _ = gu.load(name)
    model = _build(cfg, dtype)
    crit = _crit()
    data = _data(cfg, B)
    o0, g0 = _step(model, crit, data, regions=False)
    ops.counters(reset=True)
    o1, g1 = _step(model, crit, data, regions=True)
    cnt = ops.counters()
    _assert_bitwise(o0, o1, f"{name} {dtype} outputs")
    # (bf16: the fp32 gradients of the composite matrices are rounded to bf16 before the products that recover to_q / to_k / to_v / to_out from them:
    #  a last-bit difference in fp32 can flip such a rounding)
    _assert_bitwise(g0, g1, f"{name} {dtype} gradients", order_tol=1e-5 if dtype == "fp32" else 5e-3)
    if dtype == "bf16" and (B * cfg.num_patches) % 256 == 0:            # the measured kernels served the fused regions too
        assert cnt["gemm256p"] + cnt["gemm256"] >= 12 * 12 and cnt["mhsa_bwd_bf16"] == cfg.depth, cnt


def test_q_prescale_policy_both_paths_both_settings():
    """host option attn_qpre (default 1: bf16 encoder blocks run their qkv GEMM on a q-scaled weight copy and the attention kernels with DEVIAS_ATTN_Q_PRESCALED, ABI 167):
    with the option off the fused regions and the per-kernel path are bitwise equal too (the unscaled form of rounds 1-5), the launch counter says which form ran, and the
    two settings agree to bf16 rounding (the scaled copy rounds q * c once where the unscaled form rounds q, then q * c)."""
    from devias_amd import ops
    fx, cfg, _ = gu.load("vits_t8")
    model = _build(cfg, "bf16")
    crit = _crit()
    data = _data(cfg, 2)
    res = {}
    try:
        for qpre in (1, 0):
            ops.set_option("attn_qpre", qpre)
            o0, g0 = _step(model, crit, data, regions=False)
            ops.counters(reset=True)
            o1, g1 = _step(model, crit, data, regions=True)
            cnt = ops.counters()
            assert cnt["mhsa_qpre"] == (2 * cfg.depth if qpre else 0), (qpre, cnt)
            _assert_bitwise(o0, o1, f"attn_qpre = {qpre}: outputs")
            _assert_bitwise(g0, g1, f"attn_qpre = {qpre}: gradients", order_tol=5e-3)
            res[qpre] = (o1, g1)
    finally:
        ops.set_option("attn_qpre", 1)
    for k in ("slots_head", "slots", "mask"):
        a, b = res[1][0][k].float(), res[0][0][k].float()
        assert float((a - b).abs().max() / b.abs().max()) < 3e-2, k
    assert abs(float(res[1][0]["total"]) - float(res[0][0]["total"])) / abs(float(res[0][0]["total"])) < 2e-3


@pytest.mark.parametrize("name,dtype,B", [("vits_t8", "fp32", 2), ("vits_t8", "bf16", 2), ("vitb_t16", "bf16", 8)])
def test_regions_defer_is_bitwise_the_per_kernel_second_stages(name, dtype, B):
    """`regions_defer` (default 1): an encoder block's backward runs the second stages of its partial reductions (LayerNorm parameter gradients, bias-gradient
    column sums, the dfc2 epilogue's column sums) as ONE launch of the multi-job kernel at its end instead of one launch per producer -- same loops, same order:
    outputs and every gradient bitwise equal to regions_defer = 0 (the option shipped without this comparison)."""
    from devias_amd import ops
    fx, cfg, _ = gu.load(name)
    model = _build(cfg, dtype)
    crit = _crit()
    data = _data(cfg, B)
    old = ops.get_option("regions_defer")
    try:
        ops.set_option("regions_defer", 0)
        o0, g0 = _step(model, crit, data, regions=True)
        ops.set_option("regions_defer", 1)
        o1, g1 = _step(model, crit, data, regions=True)
    finally:
        ops.set_option("regions_defer", old)
    _assert_bitwise(o0, o1, f"{name} {dtype} outputs, regions_defer 0 vs 1")
    _assert_bitwise(g0, g1, f"{name} {dtype} gradients, regions_defer 0 vs 1")


def test_regions_with_stochastic_depth_same_masks():
    fx, cfg, B = gu.load("vits_t8")
    model = _build(cfg, "bf16", drop_path_rate=0.3)
    crit = _crit()
    data = _data(cfg, 4)
    o0, g0 = _step(model, crit, data, regions=False, seed=7)
    o1, g1 = _step(model, crit, data, regions=True, seed=7)
    _assert_bitwise(o0, o1, "drop-path outputs")
    _assert_bitwise(g0, g1, "drop-path gradients", order_tol=5e-3)


def test_regions_accumulate_into_existing_grads():
    """second backward without zeroing: autograd adds the region's fresh gradient tensors into .grad (update_freq > 1)"""
    fx, cfg, B = gu.load("vits_t8")
    model = _build(cfg, "fp32")
    crit = _crit()
    data = _data(cfg, 2)
    _step(model, crit, data, regions=False)
    _, g0 = _step(model, crit, data, regions=False, zero=False)
    _step(model, crit, data, regions=True)
    _, g1 = _step(model, crit, data, regions=True, zero=False)
    _assert_bitwise(g0, g1, "accumulated gradients", order_tol=1e-5)


def test_regions_write_into_gradient_buckets_without_copies():
    """with GradSync attached (world 1) the region's weight-gradient kernels write straight into the flat bucket: inside the
    post-accumulate hook the gradient of every encoder-block weight already lives at its bucket address (nothing was cloned or packed)"""
    from devias_amd.parallel import GradSync
    fx, cfg, B = gu.load("vits_t8")
    model = _build(cfg, "bf16")
    crit = _crit()
    data = _data(cfg, 2)
    _, g0 = _step(model, crit, data, regions=True)
    sync = GradSync(model)
    seen = {}
    names = {p: n for n, p in model.named_parameters()}
    orig = sync._on_grad

    def spy(p):
        seen[names[p]] = p.grad.data_ptr() == sync._view[p].data_ptr()
        orig(p)
    for h in sync._hooks:
        h.remove()
    sync._hooks = [p.register_post_accumulate_grad_hook(spy) for p in sync.params]
    _, g1 = _step(model, crit, data, regions=True)
    sync.finish()
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    sync.remove()
    _assert_bitwise(g0, g1, "bucketed gradients")
    direct = [n for n, ok in seen.items() if ok]
    weights = [n for n in seen if n.startswith("blocks.") and n.endswith(".weight")]
    assert all(seen[n] for n in weights), [n for n in weights if not seen[n]][:5]
    assert len(direct) >= 0.8 * len([n for n in seen if n.startswith("blocks.")]), (len(direct), len(seen))
