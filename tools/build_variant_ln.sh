#!/bin/bash
# build a variant of libdevias_amd.so into tools/exp/libdevias_amd_<tag>.so with extra -D flags on layernorm.hip only: tools/build_variant_ln.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Wno-unused-result -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $FLAGS "$@" -c devias_amd/csrc/layernorm.hip -o tools/exp/ln_$tag.o
objs=""
for f in api elementwise gemm attention slot_attn loss fame regions probe attn_bwd1w; do objs="$objs devias_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libdevias_amd_$tag.so tools/exp/ln_$tag.o $objs
rm -f tools/exp/ln_$tag.o
echo built tools/exp/libdevias_amd_$tag.so
