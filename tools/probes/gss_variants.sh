#!/bin/bash
# Ablation builds of csrc/gemm_ss.hip (timing only, results wrong): variants/lib_gss_<name>.so, loaded by the probes through PRV2_HIP_LIB
# (PRV2_DISPATCH=ctypes) -- the in-tree library is never touched.   usage (CPU container): bash tools/probes/gss_variants.sh
set -e
CS=patchrefinerv2_amd/csrc
mkdir -p variants
build() {  # name, defines...
  n=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc "$@" -c $CS/gemm_ss.hip -o /tmp/gss_$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_gss_$n.so $(ls $CS/*.o | grep -v gemm_ss.o) /tmp/gss_$n.o
  echo built variants/lib_gss_$n.so
}
build noepi -DPRV2_GSS_NOEPI &
build nobar -DPRV2_GSS_NOBAR &
build noepi_nodma -DPRV2_GSS_NOEPI -DPRV2_GSS_NODMA &
wait
build bare -DPRV2_GSS_NOEPI -DPRV2_GSS_NODMA -DPRV2_GSS_NOBAR &
build nomma -DPRV2_GSS_NOEPI -DPRV2_GSS_NOMMA &
wait
wait
wait
