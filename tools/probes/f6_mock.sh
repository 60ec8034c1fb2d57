#!/bin/bash
# TIMING ONLY: the fp16 + fp6 instruction mix (16 f16 MFMAs per tap + 16 fp6 K=128 MFMAs every other tap) on the bf16x3 kernel's operand traffic
set -e
cd "$(dirname "$0")/../.."
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/libprv2_hip.so.keep6
trap 'cp /tmp/libprv2_hip.so.keep6 patchrefinerv2_amd/libprv2_hip.so' EXIT
echo "== shipped"; python tools/probes/c256_bench.py
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DPRV2_F6_MOCK -c $CS/conv3x3_gate.hip -o /tmp/gate_mock.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $(ls $CS/*.o | grep -v conv3x3_gate.o) /tmp/gate_mock.o
echo "== mock"; python tools/probes/c256_bench.py
