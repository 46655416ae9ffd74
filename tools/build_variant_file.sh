#!/bin/bash
# build a variant of libdevias_amd.so into tools/exp/libdevias_amd_<tag>.so with extra flags on ONE source file: tools/build_variant_file.sh <tag> <file stem> <flags...>
# (the other objects are the regular build's: run `python -m devias_amd.build` first; the source list and per-file flags are devias_amd/build.py's)
set -e
cd "$(dirname "$0")/.."
tag=$1; stem=$2; shift; shift
FLAGS=$(python3 -c "from devias_amd import build; print(' '.join(build._flags('$stem.hip')))")
/opt/rocm/bin/hipcc $FLAGS "$@" -Iinclude -c devias_amd/csrc/$stem.hip -o tools/exp/${stem}_$tag.o
objs=""
for f in $(python3 -c "from devias_amd import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do [ $f = $stem ] && objs="$objs tools/exp/${stem}_$tag.o" || objs="$objs devias_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libdevias_amd_$tag.so $objs
rm -f tools/exp/${stem}_$tag.o
echo built tools/exp/libdevias_amd_$tag.so
