#!/bin/bash
# A/B builds of the library on the conv microbench (same box): bash tools/ab_conv.sh <variant> ...  (variants/lib_<v>.so)
# SHAPES=0,1 restricts the shape list; the in-tree library is restored afterwards.
# A variant is the library built with extra defines (the PRV2_ABL_* ablations of csrc/conv3x3_m16.hip, -DPRV2_NO_BN32 ...), e.g.
#   mkdir -p variants && make -C patchrefinerv2_amd/csrc clean && make -C patchrefinerv2_amd/csrc ../libprv2_hip.so CXXFLAGS="-O3 -std=c++17 -fPIC \
#     --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DPRV2_ABL_NOA" && cp patchrefinerv2_amd/libprv2_hip.so variants/lib_noa.so
# (variants/ is not tracked; rebuild the plain library afterwards).  tools/pmc_conv.sh and tools/probes/halo_phase_stamps.py use the same files.
cp patchrefinerv2_amd/libprv2_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp variants/lib_$v.so patchrefinerv2_amd/libprv2_hip.so
  echo "== $v"; python tools/bench_conv.py bf16x3 27 $SHAPES 2>&1 | grep -v amdgpu | grep "halo" | head -${HEAD:-8}
done
cp /tmp/lib_keep.so patchrefinerv2_amd/libprv2_hip.so
