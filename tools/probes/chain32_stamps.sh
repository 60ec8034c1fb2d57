# chain32 phase stamps: a -DC32_STAMPS build linked to /tmp and loaded through PRV2_HIP_LIB -- the in-tree library is never touched
set -e
CS=patchrefinerv2_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc -DC32_STAMPS -c $CS/chain32.hip -o /tmp/c32s.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprv2_c32s.so $(ls $CS/*.o | grep -v chain32.o) /tmp/c32s.o
PRV2_HIP_LIB=/tmp/libprv2_c32s.so PRV2_DISPATCH=ctypes C32_STAMPS=1 python tools/bench_chain32.py 2>&1 | grep -v amdgpu.ids
