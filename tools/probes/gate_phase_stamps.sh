#!/bin/bash
# Per-workgroup phase times (s_memtime) of the fused GatedConvUnit-tail kernel: builds a stamped variant of the library into /tmp
# on the GPU box and runs tools/probes/gate_phase_stamps.py against it.   bash tools/probes/gate_phase_stamps.sh
set -e
cd "$(dirname "$0")/../../patchrefinerv2_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DPRV2_GATE_STAMPS $EXTRA -c conv3x3_gate.hip -o /tmp/conv3x3_gate_st.o
objs=$(ls *.o | grep -v conv3x3_gate.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprv2_hip_stamps.so $objs /tmp/conv3x3_gate_st.o
cd ../..
PRV2_DISPATCH=ctypes PRV2_LIB_OVERRIDE=/tmp/libprv2_hip_stamps.so python tools/probes/gate_phase_stamps.py  # (ctypes route: the torch-op library links the in-tree build)
