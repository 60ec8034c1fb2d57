#!/bin/bash
# registers / scratch of the 256-column conv kernels (conv3x3_gate.hip incl. conv3x3_w4.h), and the w4 kernel's ISA in /tmp/w4t/w4.s
mkdir -p /tmp/w4t
cd "$(dirname "$0")/../../patchrefinerv2_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -Rpass-analysis=kernel-resource-usage $EXTRA -c conv3x3_gate.hip -o conv3x3_gate.o 2>&1 |
  grep -E "error|Function Name|VGPRs:|ScratchSize" | paste - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[a-z0-9_./]*\.[a-z]*:[0-9:]* remark://g' | grep -E "${1:-w4}|error"
if [ -n "$2" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc $EXTRA -S --cuda-device-only -o /tmp/w4t/gate.s conv3x3_gate.hip
  awk '/^_ZN4prv222conv3x3_w4_gate_kernel/,/s_endpgm/' /tmp/w4t/gate.s > /tmp/w4t/w4.s
  echo "ISA lines $(wc -l < /tmp/w4t/w4.s), scratch ops $(grep -c scratch_ /tmp/w4t/w4.s), vmcnt(0) waits $(grep -c 'vmcnt(0)' /tmp/w4t/w4.s)"
fi
