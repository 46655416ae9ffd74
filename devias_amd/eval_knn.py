"""k-NN evaluation surfaces of the reference (utils/eval/run_knn.py): feature export with the model's eval forward, and the
temperature-weighted k-NN vote.  The similarity matrix is the one heavy operation and runs on the HIP GEMM (fp32 mode, exact
fp32 MFMA); top-k and the vote are [chunk, k] bookkeeping on torch ops.  Datasets / loaders are the caller's (out of scope)."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


@torch.no_grad()
def extract_features(model, scene_model, data_loader, num_samples: int):
    """run_knn.py:28-120.  Batches are (clips, label, index); returns (action_features [n, D], scene_features [n, D],
    scene_targets [n]) on rank 0 (None elsewhere), rows placed by dataset index.  `scene_targets` is the argmax of the frozen
    teacher's scene logits, as in the reference."""
    model.eval()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    action_features = scene_features = scene_targets = None
    for batch in data_loader:
        samples = batch[0].cuda(non_blocking=True)
        index = batch[-1].cuda(non_blocking=True)
        (action_feats, scene_feats), _, _ = model(samples)
        _, teacher_scene_logit = scene_model(samples, return_attn=False)
        scene_target = torch.argmax(teacher_scene_logit.float(), dim=1).float()
        action_feats, scene_feats = action_feats.float().contiguous(), scene_feats.float().contiguous()
        if world > 1:
            def gather(t):
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                return torch.cat(parts)
            index, action_feats, scene_feats, scene_target = gather(index), gather(action_feats), gather(scene_feats), gather(scene_target)
        if rank == 0:
            if action_features is None:
                action_features = torch.zeros(num_samples, action_feats.shape[-1], device=samples.device)
                scene_features = torch.zeros(num_samples, scene_feats.shape[-1], device=samples.device)
                scene_targets = torch.zeros(num_samples, device=samples.device)
            action_features.index_copy_(0, index, action_feats)
            scene_features.index_copy_(0, index, scene_feats)
            scene_targets.index_copy_(0, index, scene_target)
    return action_features, scene_features, scene_targets


@torch.no_grad()
def knn_classifier(train_features, train_labels, test_features, test_labels, k, T, num_classes=1000, num_chunks=100):
    """run_knn.py:123-163: for each test feature the k most similar training features (dot product) vote for their label with
    weight exp(similarity / T); returns (top1, top5) in percent.  Features [n, D] fp32 on the GPU, labels int64."""
    train_features = train_features.float().contiguous()
    test_features = test_features.float().contiguous()
    n_test = test_labels.shape[0]
    per = max(1, n_test // num_chunks)
    top1 = top5 = total = 0
    for i in range(0, n_test, per):
        feats = test_features[i:i + per].contiguous()
        targets = test_labels[i:i + per]
        similarity = ops.gemm(feats, train_features)                        # [chunk, n_train] = feats @ train^T, fp32 MFMA
        distances, indices = similarity.topk(k, largest=True, sorted=True)
        neighbours = train_labels.view(1, -1).expand(feats.shape[0], -1).gather(1, indices)
        votes = torch.zeros(feats.shape[0], num_classes, device=feats.device)
        votes.scatter_add_(1, neighbours, (distances / T).exp())
        _, predictions = votes.sort(dim=1, descending=True, stable=True)      # ties (classes without votes) in class order, as the CPU sort of the reference run
        correct = predictions.eq(targets.view(-1, 1))
        top1 += int(correct[:, :1].sum())
        top5 += int(correct[:, :min(5, k)].sum())
        total += targets.shape[0]
    return top1 * 100.0 / total, top5 * 100.0 / total
