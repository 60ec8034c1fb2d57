#!/bin/bash
# Same-box A/B of library builds on any probe:  bash tools/ab_probe.sh "<probe command>" <variant> ...   (variants/lib_<v>.so;
# "tree" = the in-tree library).  Two interleaved rounds (the chip's clock wanders; cdna guide rule 24); the in-tree library is
# restored afterwards.  See tools/ab_conv.sh for how a variant library is built.
PROBE="$1"; shift
cp patchrefinerv2_amd/libprv2_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so patchrefinerv2_amd/libprv2_hip.so' EXIT  # also when interrupted: never leave a variant library in the tree
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = tree ]; then cp /tmp/lib_keep.so patchrefinerv2_amd/libprv2_hip.so; else cp variants/lib_$v.so patchrefinerv2_amd/libprv2_hip.so; fi
    echo "== $v (round $round)"; $PROBE 2>&1 | grep -v amdgpu.ids
  done
done
