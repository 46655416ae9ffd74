"""Can two ranks share ONE GPU under RCCL on this image?  (the pool has 1-GPU boxes; a yes would let the N > 1 path run over the real backend)
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/exp/rccl_same_gpu.py"""
import os, torch, torch.distributed as dist
r = int(os.environ["RANK"]); w = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=r, world_size=w, device_id=torch.device("cuda", 0))
x = torch.full((1 << 20,), float(r + 1), device="cuda", dtype=torch.bfloat16)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rank {r}: all_reduce ok, value {x[0].item()} (expected {w * (w + 1) / 2})", flush=True)
dist.destroy_process_group()
