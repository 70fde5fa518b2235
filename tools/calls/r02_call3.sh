#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call3
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "not bench_path and not ntxent and not lars_vs" > "$OUT/pytest_rest.log" 2>&1
tail -8 "$OUT/pytest_rest.log"; grep -n "fixed_\|grad-error share\|FAILED\|Error" "$OUT/pytest_rest.log" | head -90
