"""Which parameter gradients does GradSync still have to copy into its buckets (they did not arrive as the bucket view)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
sys.argv = [sys.argv[0], "--batch", "4"]
args = bench.parse()
dev = torch.device("cuda", 0)
from devias_amd import synth
from devias_amd.parallel import GradSync
from devias_amd.train_loss import TrainLoss
model = bench.build_model(args, dev)
names = {p: n for n, p in model.named_parameters()}
sync = GradSync(model, simulate=True)
copied = []
orig = sync._on_grad
def spy(p):
    if p.grad.data_ptr() != sync._view[p].data_ptr():
        copied.append((names[p], tuple(p.shape)))
    return orig(p)
for h in sync._hooks: h.remove()
sync._hooks = [p.register_post_accumulate_grad_hook(spy) for p in sync.params]
B = 4; N = model.patch_embed.num_patches
x = synth.video(B, 16, 224, seed=1000).to(dev); y = synth.targets(B, 400, seed=1000).to(dev); tl = synth.teacher_logits(B, 365, seed=1000).to(dev)
fg = tuple(t.to(dev) for t in synth.fg_masks(B, N, 196, seed=1000))
crit = TrainLoss(scene_criterion="KL", num_action_classes=400, slot_matching_method="matching", scene_loss_weight=4000, mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0, sync_loss_dict=False)
for step in range(2):
    copied.clear()
    for p in model.parameters(): p.grad = None
    out = model(x); total, _, _ = crit(model, out, (None, tl), y, fg_mask=fg); total.backward(); sync.finish()
torch.cuda.synchronize()
print(len(copied), "of", len(sync.params), "gradients were copied into their bucket view:")
for n, s in copied: print("  ", n, s)
