#!/bin/bash
# round 5, call 8: backward directions of the ResNet-50 wiring case (folded tail BatchNorm backward) against the reference's central differences
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call8
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "single_step_matches" > "$OUT/pytest.log" 2>&1
grep -n "backward_vs\|passed\|failed" "$OUT/pytest.log" | cut -c1-220 | tail -14
grep -n "'got'" "$OUT/pytest.log" | head -3 | cut -c1-600
