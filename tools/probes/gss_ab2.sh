#!/bin/bash
# GPU box: tree vs variants on the four ViT-L linears as the blocks call them (tools/probes/vit_ab.py part 1), two interleaved rounds
export PRV2_DISPATCH=ctypes
for round in 1 2; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset PRV2_HIP_LIB; else export PRV2_HIP_LIB=$(pwd)/variants/lib_gss_$v.so; fi
    echo "== $v (round $round)"; GSS_ONLY=ppb2 python tools/probes/vit_ab.py 14 1 2>&1 | grep -v amdgpu.ids | grep "persist ppb2"
  done
done
