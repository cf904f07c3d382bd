#!/bin/bash
# Build an experimental variant of libddmi.so: tools/build_variant.sh <name> "<extra -D flags>"
# -> diffdock_amd/csrc/build/var_<name>.so   (bench.py --lib that path for an A/B run on the GPU box)
# Every source is compiled with the flags (-DDDMI_PROFILING=1 switches on the DDMI_ABLATE / DDMI_FREEZE_POSE hooks, =2 also the
# in-kernel phase clocks of k_conv_fused; none of this exists in the shipped library).
# BASE=<other variant>: only the k_conv*.hip translation units (or the ones listed in ONLY="k_hidden ...") are compiled with the flags, the other objects are taken from var_<BASE>.
set -e
cd "$(dirname "$0")/../diffdock_amd/csrc"
name=$1; shift
mkdir -p build/var_${name}
pids=""
cc() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@"; }
if [ -n "$BASE" ]; then
  cp build/var_${BASE}/*.o build/var_${name}/
  for f in ${ONLY:-k_conv k_conv_f32 k_conv_bf k_conv_l2 k_conv_grp}; do
    cc $@ -x hip -c $f.hip -o build/var_${name}/$f.o &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
else
  for f in k_gemm k_conv k_conv_f32 k_conv_bf k_conv_l2 k_conv_grp k_hidden k_node k_graph k_embed k_readout k_sample; do
    cc $@ -x hip -c $f.hip -o build/var_${name}/$f.o &
    pids="$pids $!"
  done
  for f in o3_host weights complex api; do
    cc $@ -x hip -c $f.cpp -o build/var_${name}/$f.o &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var_${name}.so build/var_${name}/*.o
echo build/var_${name}.so
