#!/bin/bash
# Builds libsimclr_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
# SIMCLR_SO_OUT / SIMCLR_BUILD_DIR: build somewhere else (compile checks while a GPU call is using the in-tree library)
OUT=${SIMCLR_SO_OUT:-../libsimclr_hip.so}
B=${SIMCLR_BUILD_DIR:-build}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p $B
pids=()
for f in runtime ntxent lars conv bn pool augment comm; do
  if [ ! -f $B/$f.o ] || [ $f.hip -nt $B/$f.o ] || [ common.h -nt $B/$f.o ] || { [ $f = conv ] && [ igemm_wide.h -nt $B/$f.o ]; }; then
    hipcc $FLAGS -c $f.hip -o $B/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $B/runtime.o $B/ntxent.o $B/lars.o $B/conv.o $B/bn.o $B/pool.o $B/augment.o $B/comm.o -o $OUT
echo "built $(realpath $OUT)"
if [ "$1" = "diag" ]; then
  # diagnostic library (tools/diag_conv.py): conv kernels with run-time switches that skip pipeline parts
  hipcc $FLAGS -DSIMCLR_DIAG -c conv.hip -o $B/conv_diag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $B/runtime.o $B/ntxent.o $B/lars.o $B/conv_diag.o $B/bn.o $B/pool.o $B/augment.o $B/comm.o -o ../libsimclr_hip_diag.so
  echo "built $(realpath ../libsimclr_hip_diag.so)"
fi
