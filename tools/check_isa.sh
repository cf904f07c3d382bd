#!/bin/bash
# Device-code sanity of a built object (no GPU needed): tools/check_isa.sh [diffdock_amd/csrc/build/k_conv.o]
# Prints the number of scratch_ (register spills / stack arrays) and flat_ (address space lost) instructions of the gfx950 code object.
# k_conv.o must show < 100 scratch_ (a few spilled registers in the tile prologue of each kernel) and ~10 flat_: the round-5 "one select per row" epilogue compiled to 11 747 scratch_
# instructions and ran at 62 instead of 147 poses/s (profiles/r05_e5_ab.txt).
o=${1:-$(dirname "$0")/../diffdock_amd/csrc/build/k_conv.o}
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$o" $t/copy.o 2>/dev/null   # (an output file: without one objcopy rewrites its input)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/dev.co --unbundle 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d $t/dev.co > $t/dev.dis 2>/dev/null
echo "$o: scratch_ $(grep -c 'scratch_' $t/dev.dis)  flat_ $(grep -c 'flat_' $t/dev.dis)  v_mfma $(grep -c 'v_mfma' $t/dev.dis)"
rm -rf $t
