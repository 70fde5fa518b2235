#!/bin/bash
# round 3, call 2: parity + A/B of the nine-tap 3x3 wgrad (wave = 16 ci x 64 co x 9 taps) and of the 256x256 fwd/dgrad tile;
# trajectory sweep; the reworked soundness / depth-152 tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call2
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "multitap or 256_tile" > "$OUT/pytest_kernels.log" 2>&1
tail -3 "$OUT/pytest_kernels.log" | cut -c1-300; grep -n "FAILED\|Error\|err=" "$OUT/pytest_kernels.log" | head -30 | cut -c1-250
timeout 200 python tools/bench_wgrad3x3.py --wide > "$OUT/wgrad3x3.txt" 2>&1; cat "$OUT/wgrad3x3.txt" | grep -v amdgpu.ids
timeout 400 python tools/microbench.py --what tile --out "$OUT/tile_ab.json" > "$OUT/tile_ab.txt" 2>&1; grep -v amdgpu.ids "$OUT/tile_ab.txt"
timeout 300 python tools/traj_sweep.py "$OUT/traj_sweep.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/traj_sweep.txt"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "trained_point or resnet152" > "$OUT/pytest_new.log" 2>&1
tail -3 "$OUT/pytest_new.log" | cut -c1-300; grep -n "err=\|FAILED\|Error\|pretrained" "$OUT/pytest_new.log" | head -60 | cut -c1-220
