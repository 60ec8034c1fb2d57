#!/bin/bash
# A/B builds of the library on the conv microbench (same box): bash tools/ab_conv.sh <variant> ...  (variants/lib_<v>.so)
# SHAPES=0,1 restricts the shape list; the in-tree library is restored afterwards
cp patchrefinerv2_amd/libprv2_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp variants/lib_$v.so patchrefinerv2_amd/libprv2_hip.so
  echo "== $v"; python tools/bench_conv.py bf16x3 27 $SHAPES 2>&1 | grep -v amdgpu | grep "halo" | head -${HEAD:-8}
done
cp /tmp/lib_keep.so patchrefinerv2_amd/libprv2_hip.so
