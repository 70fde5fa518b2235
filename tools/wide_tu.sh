#!/bin/bash
# tools/wide_tu.sh: compile conv_igemm_wide alone (seconds instead of a minute) and print its registers / hot-loop statistics.
# Register-pressure experiments on csrc/igemm_wide.h; the product build stays simclr_amd/csrc/build.sh.
set -e
cd "$(dirname "$0")/../simclr_amd/csrc"
T=${TMPDIR:-/tmp}/wide_tu
mkdir -p $T
N=$(grep -n '^// forward / dgrad implicit GEMM.  Tile BM=128' conv.hip | cut -d: -f1)     # end of the shared helpers
{ sed -n "1,$((N - 2))p" conv.hip; echo '#include "igemm_wide.h"';
  echo '}';
  echo '__attribute__((used)) void wide_tu_launch(ConvP p) {';
  echo '  hipLaunchKernelGGL((conv_igemm_wide<MODE_DGRAD, true, true, true>), dim3(256), dim3(512), 153600, 0, p);';
  echo '  hipLaunchKernelGGL((conv_igemm_wide<MODE_DGRAD, true, true, false>), dim3(256), dim3(512), 153600, 0, p);';
  echo '}'; } > $T/wide_tu.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I. -c $T/wide_tu.hip -o $T/wide_tu.o
cd ../..
python tools/kernel_resources.py $T/wide_tu.o | cut -c1-150
tools/dis.sh $T/wide_tu.o $T/wide_tu.s
for k in "conv_igemm_wideILi1ELb1ELb1ELb1" "conv_igemm_wideILi1ELb1ELb1ELb0"; do python tools/isa_loops.py $T/wide_tu.s $k | grep "mfma= *[1-9]\|kernel" | head -${1:-6}; done
