#!/bin/bash
# round 6, call 9: split-fp16 NT-Xent sweeps -- oracle / fixture gates, microbench at the cfg2 / cfg3 shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call9
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ntxent" > "$OUT/pytest_ntx.txt" 2>&1; tail -15 "$OUT/pytest_ntx.txt"
timeout 300 python tools/microbench.py --what ntxent --iters 20 > "$OUT/ntxent_micro.txt" 2>&1; cat "$OUT/ntxent_micro.txt" | tail -10
