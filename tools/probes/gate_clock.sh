#!/bin/bash
# In-kernel clock and cycles per tile of the two GatedConvUnit-tail kernels on random and on all-zero operands (stamped build of the
# library in /tmp on the GPU box).   bash tools/probes/gate_clock.sh
set -e
cd "$(dirname "$0")/../../patchrefinerv2_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DPRV2_GATE_STAMPS $EXTRA -c conv3x3_gate.hip -o /tmp/conv3x3_gate_st.o
objs=$(ls *.o | grep -v conv3x3_gate.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprv2_hip_stamps.so $objs /tmp/conv3x3_gate_st.o
cd ../..
PRV2_LIB_OVERRIDE=/tmp/libprv2_hip_stamps.so python tools/probes/gate_clock.py
