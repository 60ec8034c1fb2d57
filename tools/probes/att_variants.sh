#!/bin/bash
# A/B builds of csrc/attention.hip (bit-exact switches): variants/lib_att_<name>.so, loaded through PRV2_HIP_LIB (PRV2_DISPATCH=ctypes)
set -e
CS=patchrefinerv2_amd/csrc
mkdir -p variants
build() {
  n=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc "$@" -c $CS/attention.hip -o /tmp/att_$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_att_$n.so $(ls $CS/*.o | grep -v attention.o) /tmp/att_$n.o
  echo built variants/lib_att_$n.so
}
for v in "$@"; do
  case $v in
    rescale) build rescale -DPRV2_ATT_ALWAYS_RESCALE & ;;
    setprio) build setprio -DPRV2_ATT_SETPRIO & ;;
    *) build $v -D$v & ;;
  esac
done
wait
