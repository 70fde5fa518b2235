#!/bin/bash
# tools/dis.sh <object.o> <out.s>: gfx950 disassembly of a HIP object (fat binary unbundled)
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/dev.co > "$2"
rm -rf $T
