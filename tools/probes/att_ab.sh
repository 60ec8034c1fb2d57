#!/bin/bash
# GPU box: tree vs attention variants (tools/probes/att_variants.sh), attention block alone, 14 x 1025 and 41 x 769 tokens, two rounds
export PRV2_DISPATCH=ctypes ATT_ONLY=1
for round in 1 2; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset PRV2_HIP_LIB; else export PRV2_HIP_LIB=$(pwd)/variants/lib_att_$v.so; fi
    a=$(NTOK=1025 python tools/probes/vit_ab.py 14 3 2>&1 | grep "attention alone" | sed 's/.*split-swizzled qkv //')
    b=$(NTOK=769 python tools/probes/vit_ab.py 41 3 2>&1 | grep "attention alone" | sed 's/.*split-swizzled qkv //')
    echo "$v (round $round): 14 x 1025: $a   41 x 769: $b"
  done
done
