#!/bin/bash
# timing ablations of csrc/upconv5.hip on the GPU box (results wrong by design): main loop only / gather only
set -e
cd "$(dirname "$0")/../.."
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/libprv2_hip.so.keep5
trap 'cp /tmp/libprv2_hip.so.keep5 patchrefinerv2_amd/libprv2_hip.so' EXIT
for v in "" "-DUPC5_ABL_NOMAIN" "-DUPC5_ABL_NOWALK" "-DUPC5_ABL_NOMAIN -DUPC5_ABL_NOWALK"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc $v -c $CS/upconv5.hip -o /tmp/upconv5_v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $(ls $CS/*.o | grep -v upconv5.o) /tmp/upconv5_v.o
  echo "== variant: ${v:-shipped}"
  python tools/bench_upconv5.py 2>&1 | grep upconv5x5
done
