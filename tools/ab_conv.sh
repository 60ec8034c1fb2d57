#!/bin/bash
# A/B two builds of the library on the conv microbench (same process order, same box)
for v in "$@"; do
  cp variants/lib_$v.so patchrefinerv2_amd/libprv2_hip.so
  echo "== $v"; python tools/bench_conv.py bf16x3 27 2>&1 | grep -v amdgpu | grep "halo" | head -8
done
