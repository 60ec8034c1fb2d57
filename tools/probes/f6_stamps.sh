#!/bin/bash
# phase stamps of csrc/conv3x3_f6.hip on the GPU box; extra -D flags as arguments
cd "$(dirname "$0")/../.."
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/libprv2_hip.so.keep8
trap 'cp /tmp/libprv2_hip.so.keep8 patchrefinerv2_amd/libprv2_hip.so' EXIT
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DF6_STAMPS $v -c $CS/conv3x3_f6.hip -o /tmp/f6_s.o 2>&1 | grep -i error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $(ls $CS/*.o | grep -v conv3x3_f6.o) /tmp/f6_s.o
  echo "== stamps, variant: ${v:-shipped}"
  timeout 200 python tools/probes/f6_stamps.py 2>&1 | grep -v amdgpu.ids
done
