#!/bin/bash
# timing ablations of csrc/chain32.hip on the GPU box: rebuilds the library with -DC32_ABL_* and runs tools/bench_chain32.py (results wrong by design)
set -e
cd "$(dirname "$0")/.."
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/libprv2_hip.so.keep
trap 'cp /tmp/libprv2_hip.so.keep patchrefinerv2_amd/libprv2_hip.so' EXIT
for v in "" "-DC32_ABL_NOE1" "-DC32_ABL_NOE2" "-DC32_ABL_NOSTORE" "-DC32_ABL_NOWIN" "-DC32_ABL_NOE1 -DC32_ABL_NOE2 -DC32_ABL_NOSTORE -DC32_ABL_NOWIN" $EXTRA_VARIANTS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc $v -c $CS/chain32.hip -o /tmp/chain32_v.o 2>/dev/null
  objs=$(ls $CS/*.o | grep -v chain32.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $objs /tmp/chain32_v.o
  echo "== variant: ${v:-shipped}"
  python tools/bench_chain32.py 2>&1 | grep chain32_
done
