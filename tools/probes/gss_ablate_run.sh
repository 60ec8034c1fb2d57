#!/bin/bash
# GPU box: time the gemm_ss ablation builds (tools/probes/gss_variants.sh) on the ViT-L linear shapes, two interleaved rounds
export PRV2_DISPATCH=ctypes PRV2_GEMM_SS_TILE=256
for round in 1 2; do
  for v in tree noepi nobar noepi_nodma bare; do
    if [ $v = tree ]; then unset PRV2_HIP_LIB; else export PRV2_HIP_LIB=$(pwd)/variants/lib_gss_$v.so; fi
    echo "== $v ppb${PRV2_GSS_PPB:-2} (round $round)"; python tools/probes/gemm_ss_time.py 2>&1 | grep -v amdgpu.ids
  done
  export PRV2_HIP_LIB=$(pwd)/variants/lib_gss_nomma.so
  echo "== nomma ppb8 (round $round)"; PRV2_GSS_PPB=8 python tools/probes/gemm_ss_time.py 2>&1 | grep -v amdgpu.ids
done
