#!/bin/bash
# Build an experimental variant of libddmi.so: tools/build_variant.sh <name> "<extra -D flags>"
# -> diffdock_amd/csrc/build/var_<name>.so   (bench.py --lib that path for an A/B run on the GPU box)
set -e
cd "$(dirname "$0")/../diffdock_amd/csrc"
make -j8 >/dev/null
name=$1; shift
for f in k_conv k_embed; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@ -x hip -c $f.hip -o build/var_${name}_$f.o
done
objs=$(ls build/*.o | grep -v "emu_\|var_\|k_conv.o\|k_embed.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var_${name}.so $objs build/var_${name}_k_conv.o build/var_${name}_k_embed.o
echo build/var_${name}.so
