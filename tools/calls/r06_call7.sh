#!/bin/bash
# round 6, call 7: per-layer table of the parity mode (f16x3_3, pre-split gradients) + adam test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call7
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_optimizers.py -q -m gpu > "$OUT/pytest_opt.txt" 2>&1; tail -3 "$OUT/pytest_opt.txt"
timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_parity.txt" 2>&1
cat "$OUT/per_layer_parity.txt" | tail -30
