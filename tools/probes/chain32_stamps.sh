set -e
CS=patchrefinerv2_amd/csrc
cp patchrefinerv2_amd/libprv2_hip.so /tmp/keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DC32_STAMPS -c $CS/chain32.hip -o /tmp/c32s.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o patchrefinerv2_amd/libprv2_hip.so $(ls $CS/*.o | grep -v chain32.o) /tmp/c32s.o
C32_STAMPS=1 python tools/bench_chain32.py 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so patchrefinerv2_amd/libprv2_hip.so
