#!/bin/bash
# Device-code gate of a built object (no GPU needed): tools/check_isa.sh [object ...]   (default: every k_conv*.o of the build)
# Counts, in the gfx950 code object: scratch_ (register spills / stack arrays), flat_ (address space lost), v_mfma_f32_16x16x4_f32
# (the exact-f32 matrix instructions of the convolution), the largest vgpr_spill_count / sgpr_spill_count of the kernel
# descriptors -- and FAILS (exit 1) when a count is on the wrong side of the committed limits in tools/isa_limits.txt
# (lines: <object basename> <max scratch_> <max flat_> <min v_mfma_f32_16x16x4_f32> <max vgpr_spill_count>).
# Why a gate: the round-5 "one select per row" epilogue compiled to 11 747 scratch_ instructions and ran at 62 instead of 147
# poses/s (profiles/r05_e5_ab.txt); nothing but a disassembly shows that before a GPU run.  __graft_entry__.build() runs this.
here=$(cd "$(dirname "$0")" && pwd)
limits=$here/isa_limits.txt
objs=("$@")
if [ ${#objs[@]} -eq 0 ]; then objs=($here/../diffdock_amd/csrc/build/k_conv*.o); fi
LLVM=/opt/rocm/lib/llvm/bin
rc=0
for o in "${objs[@]}"; do
  case "$(basename $o)" in emu_*) continue;; esac
  t=$(mktemp -d)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$o" $t/copy.o 2>/dev/null   # (an output file: without one objcopy rewrites its input)
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/dev.co --unbundle 2>/dev/null
  $LLVM/llvm-objdump -d $t/dev.co > $t/dev.dis 2>/dev/null
  $LLVM/llvm-readelf --notes $t/dev.co > $t/notes.txt 2>/dev/null
  scratch=$(grep -c 'scratch_' $t/dev.dis); flat=$(grep -c 'flat_' $t/dev.dis); mfma=$(grep -c 'v_mfma' $t/dev.dis)
  mfma_f32=$(grep -c 'v_mfma_f32_16x16x4_f32' $t/dev.dis)
  vspill=$(grep -o 'vgpr_spill_count: *[0-9]*' $t/notes.txt | awk '{print $2}' | sort -n | tail -1); vspill=${vspill:-0}
  sspill=$(grep -o 'sgpr_spill_count: *[0-9]*' $t/notes.txt | awk '{print $2}' | sort -n | tail -1); sspill=${sspill:-0}
  echo "$o: scratch_ $scratch  flat_ $flat  v_mfma $mfma  v_mfma_f32_16x16x4_f32 $mfma_f32  max vgpr_spill_count $vspill  max sgpr_spill_count $sspill"
  rm -rf $t
  b=$(basename $o)
  if [ -f $limits ]; then
    read -r _ max_scratch max_flat min_mfma max_vspill <<< "$(grep "^$b " $limits | head -1)"
    if [ -n "$max_scratch" ]; then
      if [ "$scratch" -gt "$max_scratch" ]; then echo "  FAIL: scratch_ $scratch > $max_scratch"; rc=1; fi
      if [ "$flat" -gt "$max_flat" ]; then echo "  FAIL: flat_ $flat > $max_flat"; rc=1; fi
      if [ "$mfma_f32" -lt "$min_mfma" ]; then echo "  FAIL: v_mfma_f32_16x16x4_f32 $mfma_f32 < $min_mfma"; rc=1; fi
      if [ "$vspill" -gt "$max_vspill" ]; then echo "  FAIL: vgpr_spill_count $vspill > $max_vspill"; rc=1; fi
    fi
  fi
done
exit $rc
