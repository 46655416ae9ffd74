#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4rccl; mkdir -p $O
for envs in "A=1"; do
  echo "=== env: $envs"
  env $envs NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/exp/rccl_same_gpu.py 2>&1 | grep -i "error\|duplicate\|ok, value\|invalid" | grep -v "error_file\|elastic/errors\|Could not read\|iommu" | cut -c1-300 | head -12
done > $O/probe.txt 2>&1
cat $O/probe.txt
