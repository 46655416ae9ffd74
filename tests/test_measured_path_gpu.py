"""Parity ON THE MEASURED KERNELS (VERDICT r1 "weak" 1-2, "next" 1, 6, 9): bf16 end-to-end steps whose token matrix is a multiple of
256 rows, so that the forward AND dgrad GEMMs run in the 256x256 LDS-DMA kernels (persistent form included) and attention in the MFMA
kernels -- asserted through the library's launch counters -- compared with the reference golden (chunk 0) and the CPU oracle evaluated
chunk by chunk with per-chunk pad-min (SURVEY.md §8e composition rule, utils/loss/train_loss.py:103-106); full-size B = 32 properties;
BASELINE configs 4 and 5 at their real geometry; attention at 6400 / 6401 tokens; the teacher at 1569 tokens."""
import numpy as np
import pytest
import torch

import golden_util as gu
from devias_amd import synth
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

# bf16 bounds = ~2x what was measured on MI355X in round 2 at these shapes (logits 6.3-8.1e-3, composed / total loss 0.5-1.3e-4, gradient norms
# median 1.8e-4, p90 7e-4, worst single parameter 1.7e-2)
TOL_BF16_LOGITS = 1.5e-2
TOL_BF16_LOSS = 5e-4
TOL_BF16_GRADNORM_MEDIAN = 2e-3


def _build(cfg, dtype, **kw):
    from functools import partial
    from devias_amd.modeling_slot import VisionTransformer
    m = VisionTransformer(img_size=cfg.img_size, patch_size=16, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4,
                          qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=cfg.num_classes,
                          all_frames=cfg.all_frames, tubelet_size=cfg.tubelet_size, init_scale=1e-3,
                          num_latents=cfg.num_latents, head_type=cfg.head_type, slot_matching_method="matching",
                          agg_weights_tie=cfg.agg_weights_tie, agg_depth=cfg.agg_depth,
                          num_scene_classes=cfg.num_scene_classes, compute_dtype=dtype, **kw)
    synth.fill_module_(m, seed=0)
    return m.cuda().train()


def _crit():
    from devias_amd.train_loss import TrainLoss
    return TrainLoss(criterion=None, scene_criterion="KL", num_action_classes=400, slot_matching_method="matching",
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0, scene_loss_weight=4000)


def _chunked_loss(crit, model, out, tl, y, fg, S, chunk):
    """TrainLoss applied to `chunk`-clip slices of one forward (each slice with its own teacher pad-min, exactly what `chunk`-clip
    ranks would compute), averaged: the single-process comparator of SURVEY.md §8e."""
    (_, _), (_, _, attn), (slots_head, slots, maskp) = out
    B = y.shape[0]
    nh = attn.shape[0] // B
    total, logits, lds = 0, [], []
    for c in range(0, B, chunk):
        sl = slice(c * S, (c + chunk) * S)
        o = ((None, None), (None, None, attn[c * nh:(c + chunk) * nh]), (slots_head[sl], slots[sl], maskp[sl]))
        t, lg, ld = crit(model, o, (None, tl[c:c + chunk]), y[c:c + chunk], fg_mask=(fg[0][c:c + chunk], fg[1][c:c + chunk]))
        total = total + t / (B // chunk)
        logits.append(lg); lds.append(ld)
    return total, torch.cat(logits), lds


@pytest.mark.parametrize("name,B", [("vitb_t16", 8), ("vitb_t8", 16)])
def test_bf16_step_on_full_tile_kernels_vs_reference_chunks(name, B):
    """M = B*N = 12544 = 49 x 256 rows: every encoder GEMM (forward, dgrad, wgrad) is served by the 256x256 kernels and attention by the
    MFMA kernels (asserted).  Per-slot logits of every 2-clip chunk, the chunk-composed loss and every gradient norm are compared with
    the reference golden (chunk 0: same inputs as the fixture) and with the CPU oracle run chunk by chunk."""
    from devias_amd import ops
    fx, cfg, Bg = gu.load(name)
    assert Bg == 2 and (B * cfg.num_patches) % 256 == 0
    model = _build(cfg, "bf16")
    crit = _crit()
    x = synth.video(B, cfg.all_frames, cfg.img_size, seed=1000)
    y = synth.targets(B, cfg.num_classes, seed=1000)
    tl = synth.teacher_logits(B, cfg.num_scene_classes, seed=1000)
    fg = synth.fg_masks(B, cfg.num_patches, cfg.grid * cfg.grid, seed=1000)
    ops.counters(reset=True)
    out = model(x.cuda())
    total, logits, lds = _chunked_loss(crit, model, out, tl.cuda(), y.cuda(), (fg[0].cuda(), fg[1].cuda()), cfg.num_latents, 2)
    model.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    cnt = ops.counters()
    # which kernels ran, per encoder block: qkv / fc1 forward and the fc2 dgrad (49 x {9, 12} tiles > 256 CUs) -> persistent 256^2 kernel; proj / fc2
    # forward, the other dgrads (147 tiles) and the four wgrads (split-K) -> one-tile-per-workgroup 256^2 kernel
    assert cnt["gemm256p"] >= 12 * 3 and cnt["gemm256"] >= 12 * 9 and cnt["gemm256p"] + cnt["gemm256"] >= 12 * 12, cnt
    assert cnt["mhsa_fwd_bf16"] == cfg.depth and cnt["mhsa_bwd_bf16"] == cfg.depth and cnt["mhsa_fwd_f32"] == 0, cnt
    assert cnt["gemm128_f32"] == 0, cnt
    S = cfg.num_latents
    sh = out[2][0].detach().float().cpu()
    # chunk 0 == the golden's inputs (formula data is indexed by clip): the REFERENCE's own numbers
    e0 = gu.rel(sh[:2 * S], fx["slots_head"])
    assert e0 < TOL_BF16_LOGITS, e0
    assert abs(float(lds[0]["action_loss"]) - float(fx["loss_values"][list(map(str, fx["loss_names"])).index("action_loss")])) < 5e-3
    # all chunks: CPU oracle, chunk by chunk (2 clips each, own pad-min), composed as the mean
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    o_total, o_grads, e_logits = 0.0, None, []
    for c in range(0, B, 2):
        xc, yc, tlc = x[c:c + 2], y[c:c + 2], tl[c:c + 2]
        fgc = (fg[0][c:c + 2], fg[1][c:c + 2])
        t, lg, ld, g, oo, idx = ref_cpu.train_step(P, cfg, xc, yc, tlc, fgc)
        o_total += float(t) / (B // 2)
        o_grads = g if o_grads is None else {k: o_grads[k] + g[k] for k in g}
        e_logits.append(gu.rel(sh[c * S:(c + 2) * S], oo[2][0].detach()))
    e_total = abs(float(total) - o_total) / abs(o_total)
    names = [n for n, _ in model.named_parameters()]
    gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    on = np.array([float((o_grads[n] / (B // 2)).double().norm()) for n in names])
    e_gn = np.abs(gn - on) / np.maximum(on, 1e-6 * on.max())
    print(f"{name} B={B} bf16 full-tile kernels: logits rel max {max(e_logits):.3e} (chunk 0 vs golden {e0:.3e}), composed loss rel {e_total:.3e}, "
          f"grad-norm rel median {np.median(e_gn):.3e} p90 {np.quantile(e_gn, 0.9):.3e} max {e_gn.max():.3e}")
    assert max(e_logits) < TOL_BF16_LOGITS
    assert e_total < TOL_BF16_LOSS
    assert np.median(e_gn) < TOL_BF16_GRADNORM_MEDIAN and np.quantile(e_gn, 0.9) < 2e-2


def test_bf16_full_size_step_properties():
    """BASELINE config 2 as stated (ViT-B/16 16x224^2, bf16, B = 32, M = 50176): finite; bitwise run-to-run reproducible (loss and all
    186 gradients: no float atomics, fixed-order reductions, deterministic persistent-kernel hand-offs); rows of the B = 32 forward ==
    the same clips run as four B = 8 batches (a clip's arithmetic does not depend on its batch); loss and per-slot logits within the
    bf16 bound of the fp32 parity mode of the same kernels family (itself pinned to the reference at 1e-3 by the goldens)."""
    from devias_amd import ops
    cfg = ref_cpu.SlotViTConfig(all_frames=16)
    B, S = 32, cfg.num_latents
    x = synth.video(B, 16, 224, seed=1000).cuda()
    y = synth.targets(B, 400, seed=1000).cuda()
    tl = synth.teacher_logits(B, 365, seed=1000).cuda()
    fg = tuple(t.cuda() for t in synth.fg_masks(B, cfg.num_patches, 196, seed=1000))
    crit = _crit()

    def step(model, xb, yb, tlb, fgb):
        model.zero_grad(set_to_none=True)
        out = model(xb)
        total, logits, ld = crit(model, out, (None, tlb), yb, fg_mask=fgb)
        total.backward()
        return out, total, {n: p.grad.clone() for n, p in model.named_parameters()}

    model = _build(cfg, "bf16")
    ops.counters(reset=True)
    out1, t1, g1 = step(model, x, y, tl, fg)
    cnt = ops.counters()
    # M = 50176: the four forward and four dgrad GEMMs of a block run on the eight-wave persistent kernel (tail tiles split; the
    # four-wave kernel is an option); the four wgrads split K on the one-tile-per-workgroup kernel
    assert cnt["gemm256p"] >= 12 * 8 and cnt["gemm256w"] == 0 and cnt["gemm256"] >= 12 * 4 and cnt["mhsa_bwd_bf16"] == 12 and cnt["gemm128_f32"] == 0, cnt
    # the dK / dV kernel the measured step ran: the one-wave-per-SIMD kernel, one workgroup per 256-key block + the launch for the 32 last keys of every head
    assert (cnt["dkdv1w"], cnt["dkdv1w_rest"], cnt["dkdv1w_pers"], cnt["dkdv2w"]) == (12, 12, 0, 0), cnt
    assert cnt["mhsa_qpre"] == 24, cnt          # every block's attention, both directions, on q' = q * scale * log2(e) from the qkv GEMM (DEVIAS_ATTN_Q_PRESCALED)
    sh1 = out1[2][0].detach().clone()
    out2, t2, g2 = step(model, x, y, tl, fg)
    assert torch.isfinite(t1).all() and all(torch.isfinite(g).all() for g in g1.values())
    assert torch.equal(t1, t2) and torch.equal(sh1, out2[2][0])
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    with torch.no_grad():
        for c in range(0, B, 8):
            oc = model(x[c:c + 8])
            assert torch.equal(oc[2][0], sh1[c * S:(c + 8) * S]), c
    del out2, g2
    m32 = _build(cfg, "fp32")
    with torch.no_grad():
        o32 = m32(x)
        t32, _, _ = crit(m32, o32, (None, tl), y, fg_mask=fg)
    e_logit = gu.rel(sh1.float().cpu(), o32[2][0].float().cpu())
    e_total = abs(float(t1) - float(t32)) / abs(float(t32))
    print(f"ViT-B 16x224^2 B=32 bf16 vs fp32 mode: logits rel {e_logit:.3e}, total loss rel {e_total:.3e}")
    assert e_logit < TOL_BF16_LOGITS and e_total < TOL_BF16_LOSS


def test_bf16_full_size_step_beside_the_oracle():
    """The oracle BESIDE the B = 32 kernels: the full-size bf16 step (M = 50176: the persistent 256x256 kernel with split tail tiles
    serves every forward / dgrad GEMM, asserted by the launch counters) is compared with the REFERENCE golden on chunk 0 (clips 0-1 are the `vitb_t16` fixture's inputs) and with
    the CPU oracle's train_step on four more 2-clip chunks spread over the batch: per-slot logits, matched indices, and every term of the
    chunk's loss evaluated with the chunk's own teacher pad-min.  So the schedule that only exists at B = 32 is pinned to the
    reference, not to sibling kernels."""
    from devias_amd import ops
    fx, cfg, Bg = gu.load("vitb_t16")
    assert Bg == 2 and cfg.all_frames == 16
    B, S = 32, cfg.num_latents
    x = synth.video(B, 16, 224, seed=1000)
    y = synth.targets(B, 400, seed=1000)
    tl = synth.teacher_logits(B, 365, seed=1000)
    fg = synth.fg_masks(B, cfg.num_patches, 196, seed=1000)
    model = _build(cfg, "bf16")
    crit = _crit()
    ops.counters(reset=True)
    out = model(x.cuda())
    total, logits, lds = _chunked_loss(crit, model, out, tl.cuda(), y.cuda(), (fg[0].cuda(), fg[1].cuda()), S, 2)
    model.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    cnt = ops.counters()
    assert cnt["gemm256p"] >= 12 * 8 and cnt["gemm256"] >= 12 * 4 and cnt["mhsa_fwd_bf16"] == 12 and cnt["mhsa_bwd_bf16"] == 12, cnt
    sh = out[2][0].detach().float().cpu()
    # chunk 0: the reference's own numbers
    e0 = gu.rel(sh[:2 * S], fx["slots_head"])
    names = list(map(str, fx["loss_names"]))
    for k in ("action_loss", "scene_loss", "mask_prediction_loss", "mask_distill_loss", "cosine_loss"):
        ref_v = float(fx["loss_values"][names.index(k)])
        assert abs(float(lds[0][k]) - ref_v) <= 2e-2 * max(abs(ref_v), 1e-3), (k, float(lds[0][k]), ref_v)
    assert e0 < TOL_BF16_LOGITS, e0
    # four more chunks across the batch: the CPU oracle
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    errs = [e0]
    for c in (6, 14, 22, 30):
        t, lg, ld, g, oo, idx = ref_cpu.train_step(P, cfg, x[c:c + 2], y[c:c + 2], tl[c:c + 2], (fg[0][c:c + 2], fg[1][c:c + 2]))
        errs.append(gu.rel(sh[c * S:(c + 2) * S], oo[2][0].detach()))
        mine = lds[c // 2]
        for k, v in ld.items():
            assert abs(float(mine[k]) - float(v)) <= 2e-2 * max(abs(float(v)), 1e-3), (c, k, float(mine[k]), float(v))
    print(f"ViT-B 16x224^2 B=32 bf16 vs golden / oracle on 5 chunks: per-slot logits rel {[f'{e:.2e}' for e in errs]}")
    assert max(errs) < TOL_BF16_LOGITS


def test_vit_large_full_depth_fp32_vs_oracle_and_bf16_properties():
    """BASELINE config 4 geometry: ViT-L/16 (D = 1024, 24 blocks, 16 heads).  (i) fp32 mode at full depth, 4 frames (392 tokens), B = 2,
    against the CPU oracle: logits, loss, every gradient (1e-3 / 5e-3 gates); (ii) bf16 at the REAL geometry -- 24 blocks x 1568 tokens,
    B = 4 (M = 6272 = 24.5 x 256: ragged in M -> 128^2 kernels for forward, exercised on purpose) and B = 8 (full tiles): finite,
    reproducible, B = 8 rows == B = 4 rows."""
    from devias_amd import ops
    cfg = ref_cpu.SlotViTConfig(embed_dim=1024, num_heads=16, depth=24, all_frames=4)
    B = 2
    m = _build(cfg, "fp32")
    x, y, tl, fg = gu.inputs(cfg, B)
    crit = _crit()
    out = m(x.cuda())
    total, logits, ld = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda()))
    total.backward()
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    ototal, ologits, old, ograds, oout, oidx = ref_cpu.train_step(P, cfg, x, y, tl, fg)
    assert gu.rel(out[2][0].detach().cpu(), oout[2][0].detach()) < 1e-3
    assert abs(float(total) - float(ototal)) / abs(float(ototal)) < 1e-3
    gmax = max(float(g.abs().max()) for g in ograds.values())
    worst = max(float((p.grad.cpu().double() - ograds[n].double()).abs().max() / max(float(ograds[n].abs().max()), 1e-6 * gmax))
                for n, p in m.named_parameters())
    print(f"ViT-L depth 24, 392 tokens, fp32 vs oracle: worst gradient {worst:.2e}")
    assert worst < 5e-3
    del m, out, total
    torch.cuda.empty_cache()
    # (ii) real geometry, bf16
    cfg16 = ref_cpu.SlotViTConfig(embed_dim=1024, num_heads=16, depth=24, all_frames=16)
    mb = _build(cfg16, "bf16")
    xb = synth.video(8, 16, 224, seed=1000).cuda()
    yb = synth.targets(8, 400, seed=1000).cuda()
    tlb = synth.teacher_logits(8, 365, seed=1000).cuda()
    fgb = tuple(t.cuda() for t in synth.fg_masks(8, cfg16.num_patches, 196, seed=1000))
    res = []
    for rep in range(2):
        mb.zero_grad(set_to_none=True)
        ops.counters(reset=True)
        o = mb(xb)
        t, _, _ = crit(mb, o, (None, tlb), yb, fg_mask=fgb)
        t.backward()
        cnt = ops.counters()
        res.append((o[2][0].detach().clone(), t.detach().clone(), [p.grad.clone() for p in mb.parameters()]))
    # M = 12544 = 49 row tiles: the N = 3072 / 4096 shapes (qkv, fc1 forward, fc2 dgrad: 588 / 784 tiles) run persistent, the N = 1024 ones
    # (196 tiles <= 256 CUs) one tile per workgroup
    assert cnt["gemm256p"] >= 24 * 3 and cnt["gemm256p"] + cnt["gemm256"] >= 24 * 12 and cnt["mhsa_bwd_bf16"] == 24, cnt
    assert torch.isfinite(res[0][1]).all() and all(torch.isfinite(g).all() for g in res[0][2])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    with torch.no_grad():
        o4 = mb(xb[:4])                                    # M = 6272: ragged in M -> different kernels, same arithmetic per row up to MFMA order
    assert gu.rel(o4[2][0].float().cpu(), res[0][0][:4 * cfg16.num_latents].float().cpu()) < 2e-2


def test_long_sequence_6400_tokens_step():
    """BASELINE config 5 geometry: ViT-B/16 32x320^2 -> 6400 tokens, full depth, bf16, B = 2 (M = 12800 = 50 x 256): finite, reproducible,
    and per-slot logits / loss within the bf16 bound of the fp32 parity mode (depth 12, same inputs)."""
    from devias_amd import ops
    cfg = ref_cpu.SlotViTConfig(all_frames=32, img_size=320)
    assert cfg.num_patches == 6400
    B = 2
    x = synth.video(B, 32, 320, seed=1000).cuda()
    y = synth.targets(B, 400, seed=1000).cuda()
    tl = synth.teacher_logits(B, 365, seed=1000).cuda()
    fg = tuple(t.cuda() for t in synth.fg_masks(B, 6400, 400, seed=1000))
    crit = _crit()
    mb = _build(cfg, "bf16")
    res = []
    for rep in range(2):
        mb.zero_grad(set_to_none=True)
        ops.counters(reset=True)
        o = mb(x)
        t, _, _ = crit(mb, o, (None, tl), y, fg_mask=fg)
        t.backward()
        cnt = ops.counters()
        res.append((o[2][0].detach().clone(), t.detach().clone(), [p.grad.clone() for p in mb.parameters()]))
    assert cnt["mhsa_fwd_bf16"] == 12 and cnt["mhsa_bwd_bf16"] == 12 and cnt["gemm128_f32"] == 0, cnt
    assert torch.isfinite(res[0][1]).all() and all(torch.isfinite(g).all() for g in res[0][2])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    m32 = _build(cfg, "fp32")
    with torch.no_grad():
        o32 = m32(x)
        t32, _, _ = crit(m32, o32, (None, tl), y, fg_mask=fg)
    e_logit = gu.rel(res[0][0].float().cpu(), o32[2][0].float().cpu())
    e_total = abs(float(res[0][1]) - float(t32)) / abs(float(t32))
    print(f"ViT-B 32x320^2 (6400 tokens) bf16 vs fp32 mode: logits rel {e_logit:.3e}, total loss rel {e_total:.3e}")
    assert e_logit < TOL_BF16_LOGITS and e_total < 2e-3      # measured 6.9e-3 / 4.9e-4: B = 2 only (no averaging over clips), 4x longer softmax rows


def _attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-1, -2)
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * N, H * 64), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [6400, 6401])
def test_mhsa_long_sequences(dtype, N):
    """attention kernels at the config-5 sequence length and one past it (ragged last tile), forward + backward against fp32 torch"""
    from devias_amd import ops as o
    B, H, scale = 1, 1, 0.125
    g = torch.Generator(device="cpu").manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * 64, generator=g)).cuda().to(dtype)
    out, lse = o.mhsa_fwd(qkv, B, N, H, scale)
    xr = qkv.float().clone().requires_grad_(True)
    ro, rl = _attn_ref(xr, B, N, H, scale)
    assert gu.rel(out.float().cpu(), ro.detach().cpu()) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert gu.rel(lse.cpu(), rl.detach().cpu()) < (1e-5 if dtype == torch.float32 else 1e-2)
    d_o = torch.randn(B * N, H * 64, generator=g).cuda().to(dtype)
    ro.backward(d_o.float())
    dqkv = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, scale)
    gr = xr.grad.reshape(B, N, 3, H, 64)
    mine = dqkv.float().reshape(B, N, 3, H, 64)
    for w, nm in enumerate("qkv"):
        assert gu.rel(mine[:, :, w].cpu(), gr[:, :, w].cpu()) < (2e-4 if dtype == torch.float32 else 3e-2), nm
    if dtype == torch.bfloat16:                              # bitwise reproducible
        assert torch.equal(dqkv, o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, scale))


def test_teacher_forward_1569_tokens_vs_oracle():
    """the frozen teacher at its real sequence length (16 frames -> 1568 patches + cls = 1569 tokens, ragged in every tile), full width,
    against the CPU oracle of model/modeling_finetune.py:273-325: fp32 1e-3, bf16 bounded"""
    from devias_amd.modeling_finetune import vit_base_patch16_224
    cfg = ref_cpu.SlotViTConfig(all_frames=16)
    x = synth.video(1, 16, 224, seed=1000)
    P = synth.fill_params(ref_cpu.teacher_param_shapes(cfg), seed=1)
    with torch.no_grad():
        otok, ologits = ref_cpu.teacher_forward(P, cfg, x)
    for mode, tol in (("fp32", 1e-3), ("bf16", 3e-2)):
        m = vit_base_patch16_224(num_classes=365, all_frames=16, tubelet_size=2, use_mean_pooling=False, init_scale=1e-3, compute_dtype=mode)
        synth.fill_module_(m, seed=1)
        m = m.cuda().eval()
        tok, logits = m(x.cuda(), return_attn=False)
        e1, e2 = gu.rel(tok.float().cpu(), otok), gu.rel(logits.float().cpu(), ologits)
        print(f"teacher 1569 tokens {mode}: token {e1:.2e} logits {e2:.2e}")
        assert e1 < tol and e2 < tol


def test_bf16_weights_follow_the_fused_optimizer():
    """FusedAdamW updates parameters through raw pointers, which does not move Tensor._version -- the cached bf16
    weight copies must be refreshed anyway.  With every bias / LayerNorm / latent frozen, the logits can only change through the
    bf16 MATRIX copies: they must change after opt.step(), by what the fp32 parity mode (no copies) says they change."""
    from devias_amd.modeling_slot import _WCACHE
    from devias_amd.optim import FusedAdamW
    cfg = ref_cpu.SlotViTConfig(embed_dim=384, num_heads=6, depth=2, all_frames=2, agg_depth=2)
    B = 2
    x, y, tl, fg = gu.inputs(cfg, B)
    crit = _crit()
    deltas = {}
    for mode in ("bf16", "fp32"):
        m = _build(cfg, mode)
        mats = [p for n, p in m.named_parameters() if p.dim() >= 2 and "latents" not in n]
        for p in m.parameters():
            p.requires_grad_(False)
        for p in mats:
            p.requires_grad_(True)
        opt = FusedAdamW(mats, lr=1e-3, weight_decay=0.0)
        out = m(x.cuda())
        before = out[2][0].detach().float().clone()
        total, _, _ = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda()))
        total.backward()
        casts = _WCACHE.casts
        with torch.no_grad():
            same = m(x.cuda())[2][0].float()
        assert torch.equal(same, before) and _WCACHE.casts == casts          # no update -> cache hits, identical logits
        opt.step()
        with torch.no_grad():
            after = m(x.cuda())[2][0].float()
        if mode == "bf16":
            assert _WCACHE.casts >= casts + len(mats) - 2                      # every matrix was re-cast (to_k|to_v are cast as parts of a cat)
        deltas[mode] = (after - before)
        assert float(deltas[mode].abs().max()) > 1e-4, mode
    # the change seen through the bf16 copies tracks the fp32 one
    cos = torch.nn.functional.cosine_similarity(deltas["bf16"].flatten(), deltas["fp32"].flatten(), dim=0)
    assert float(cos) > 0.98, float(cos)


def test_colsum_handoff_and_gradient_destinations():
    """(i) the bias-gradient column sums ride on the gradient tensors between backward regions (no address-keyed global):
    every encoder block but the last consumer must hit; (ii) with a GradSync attached (world 1), the encoder weight gradients are
    written straight into the flat buckets (p.grad IS the bucket view, nothing packed) and equal the gradients without GradSync bitwise."""
    from devias_amd import modeling_slot as ms
    from devias_amd.parallel import GradSync
    cfg = ref_cpu.SlotViTConfig(embed_dim=384, num_heads=6, depth=3, all_frames=2, agg_depth=2)
    B = 2
    x, y, tl, fg = gu.inputs(cfg, B)
    crit = _crit()
    m = _build(cfg, "bf16")

    def run():
        for p in m.parameters():
            p.grad = None
        out = m(x.cuda())
        total, _, _ = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda()))
        total.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    ms._COLSUM_STATS.update(hit=0, miss=0)
    ref = run()
    assert ms._COLSUM_STATS["hit"] == cfg.depth + 1 and ms._COLSUM_STATS["miss"] == 0, ms._COLSUM_STATS     # 3 blocks + patch embed
    sync = GradSync(m, bucket_bytes=1 << 20)
    got = run()
    sync.finish()
    for n, p in m.named_parameters():
        assert torch.equal(got[n], ref[n]), n
        assert p.grad.data_ptr() == sync._view[p].data_ptr(), n
    direct = [n for n, p in m.named_parameters() if n.startswith("blocks.") and p.dim() == 2]
    assert len(direct) == 4 * cfg.depth
    # accumulation window of two micro-batches == sum of two single steps
    sync.set_accumulate(True)
    out = m(x.cuda()); t, _, _ = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda())); t.backward()
    sync.set_accumulate(False)
    out = m(x.cuda()); t, _, _ = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda())); t.backward()
    sync.finish()
    for n, p in m.named_parameters():
        assert gu.rel(p.grad.cpu(), (3 * ref[n]).cpu()) < 1e-5 or float(ref[n].abs().max()) == 0, n      # 1 (left in the bucket) + 2 new
    sync.remove()
