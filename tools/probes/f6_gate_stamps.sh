#!/bin/bash
# Per-workgroup phase times (s_memtime) of conv3x3_c256_gate_f6_kernel beside the bf16x3 gate kernel's (gate_phase_stamps.sh): a stamped build of
# conv3x3_f6.hip into /tmp on the GPU box.   bash tools/probes/f6_gate_stamps.sh
set -e
cd "$(dirname "$0")/../../patchrefinerv2_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DPRV2_GATE_STAMPS -c conv3x3_f6.hip -o /tmp/conv3x3_f6_st.o
objs=$(ls *.o | grep -v conv3x3_f6.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprv2_hip_stamps6.so $objs /tmp/conv3x3_f6_st.o
cd ../..
PRV2_DISPATCH=ctypes PRV2_LIB_OVERRIDE=/tmp/libprv2_hip_stamps6.so python tools/probes/f6_gate_stamps.py
