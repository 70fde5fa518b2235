#!/bin/bash
# Builds libsimclr_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libsimclr_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p build
pids=()
for f in runtime ntxent lars conv bn pool augment; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/runtime.o build/ntxent.o build/lars.o build/conv.o build/bn.o build/pool.o build/augment.o -o $OUT
echo "built $(realpath $OUT)"
if [ "$1" = "diag" ]; then
  # diagnostic library (tools/diag_conv.py): conv kernels with run-time switches that skip pipeline parts
  hipcc $FLAGS -DSIMCLR_DIAG -c conv.hip -o build/conv_diag.o
  hipcc --offload-arch=gfx950 -shared -fPIC build/runtime.o build/ntxent.o build/lars.o build/conv_diag.o build/bn.o build/pool.o build/augment.o -o ../libsimclr_hip_diag.so
  echo "built $(realpath ../libsimclr_hip_diag.so)"
fi
