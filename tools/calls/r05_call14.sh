#!/bin/bash
# round 5, call 14: new launch-policy defaults (non-temporal BatchNorm kernels, 64-wide tile for K <= 64 only, wide forward tile from 300 GFLOP):
# the whole GPU suite, then the bench lines of cfg2 (default flags: the profile line), cfg4 and cfg5
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call14
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1300 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "suite wall seconds: $((SECONDS - T0))"; tail -2 "$OUT/pytest_gpu.log" | cut -c1-200; grep -n "^FAILED\|^E  " "$OUT/pytest_gpu.log" | head -10 | cut -c1-250
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-200 "$OUT/bench.json" | tail -1
timeout 400 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --no_parity --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-200 "$OUT/bench_cfg4.json" | tail -1
timeout 700 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --no_pmc --no_parity --prof_steps 1 > "$OUT/bench_cfg5_b256.json" 2> "$OUT/bench_cfg5_b256.err"; cut -c1-200 "$OUT/bench_cfg5_b256.json" | tail -1
echo "total: $((SECONDS - T0)) s"
