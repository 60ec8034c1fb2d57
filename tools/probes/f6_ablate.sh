#!/bin/bash
# timing ablations of csrc/conv3x3_f6.hip on the GPU box (results wrong by design)
cd "$(dirname "$0")/../.."
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/libprv2_hip.so.keep7
trap 'cp /tmp/libprv2_hip.so.keep7 patchrefinerv2_amd/libprv2_hip.so' EXIT
for v in ${F6_VARIANTS:-"" "-DF6_DBG_NODMA" "-DF6_DBG_NOW" "-DF6_DBG_NOEPI" "-DF6_ABL_NOREAD" "-DF6_ABL_NOCV" "-DF6_ABL_NOREAD -DF6_DBG_NOW -DF6_ABL_NOCV -DF6_DBG_NODMA"}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc $v -c $CS/conv3x3_f6.hip -o /tmp/f6_v.o 2>&1 | grep -i error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $(ls $CS/*.o | grep -v conv3x3_f6.o) /tmp/f6_v.o
  echo "== variant: ${v:-shipped}"
  timeout 200 python tools/probes/c256_bench.py f16f6 2>&1 | grep "14x192x256\|14x96x128"
done
