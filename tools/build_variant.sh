#!/bin/bash
# Build an experimental variant of libddmi.so: tools/build_variant.sh <name> "<extra -D flags>"
# -> diffdock_amd/csrc/build/var_<name>.so   (bench.py --lib that path for an A/B run on the GPU box)
# Every source is compiled with the flags (-DDDMI_PROFILING switches on the DDMI_ABLATE / DDMI_FREEZE_POSE hooks and the
# in-kernel phase clocks of k_conv_fused, none of which exist in the shipped library).
set -e
cd "$(dirname "$0")/../diffdock_amd/csrc"
name=$1; shift
mkdir -p build/var_${name}
pids=""
for f in k_gemm k_conv k_graph k_embed k_readout k_sample; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@ -x hip -c $f.hip -o build/var_${name}/$f.o &
  pids="$pids $!"
done
for f in o3_host weights complex api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@ -x hip -c $f.cpp -o build/var_${name}/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var_${name}.so build/var_${name}/*.o
echo build/var_${name}.so
