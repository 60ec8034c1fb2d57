cp patchrefinerv2_amd/libprv2_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp variants/lib_$v.so patchrefinerv2_amd/libprv2_hip.so
  echo "== $v"; python tools/bench_conv.py bf16x3 27 $SHAPES 2>&1 | grep -v amdgpu | grep " k1 "
done
cp /tmp/lib_keep.so patchrefinerv2_amd/libprv2_hip.so
