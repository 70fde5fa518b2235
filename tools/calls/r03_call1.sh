#!/bin/bash
# round 3, call 1: the new bf16-soundness / depth-152 tests, baseline bench + microbench of the round-2 kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call1
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "randomized_bn_batch64 or trajectory or resnet152" > "$OUT/pytest_new.log" 2>&1
tail -3 "$OUT/pytest_new.log" | cut -c1-300; grep -n "err=\|FAILED\|Error" "$OUT/pytest_new.log" | head -60 | cut -c1-220
cp gpurun_out/bf16_trajectory.json "$OUT/" 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-230 "$OUT/bench.json" | tail -1
timeout 300 python tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.txt" 2>&1; tail -5 "$OUT/microbench.txt"
