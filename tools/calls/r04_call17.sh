#!/bin/bash
# round 4, call 17: pooling tests after the maxpool_bwd bound change
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call17
mkdir -p "$OUT"
cd "$R"
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pooling or stem_backward or test_stem_conv or resnet18_f32" > "$OUT/pytest.log" 2>&1; tail -2 "$OUT/pytest.log" | cut -c1-200; grep -n "^FAILED\|^E  " "$OUT"/pytest.log | head -10 | cut -c1-250
