#!/bin/bash
# build a variant of libdevias_amd.so into tools/exp/libdevias_amd_<tag>.so with extra flags on ONE source file: tools/build_variant_file.sh <tag> <file stem> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; stem=$2; shift; shift
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Wno-unused-result -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $FLAGS "$@" -Iinclude -c devias_amd/csrc/$stem.hip -o tools/exp/${stem}_$tag.o
objs=""
for f in api elementwise layernorm attention slot_attn loss fame regions gemm; do [ $f = $stem ] && objs="$objs tools/exp/${stem}_$tag.o" || objs="$objs devias_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libdevias_amd_$tag.so $objs
rm -f tools/exp/${stem}_$tag.o
echo built tools/exp/libdevias_amd_$tag.so
