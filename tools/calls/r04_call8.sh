#!/bin/bash
# round 4, call 8: wide-tile statistics from the fp32 accumulators: parity subset, then the profile set of the final build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final2
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wide or gram or bitwise or 256_tile or determinis or test_train_step_bf16 or fixed_thresholds or bench_path_shapes" > "$OUT/pytest_subset.log" 2>&1
tail -3 "$OUT/pytest_subset.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_subset.log" | head -20 | cut -c1-250
bash tools/gpu_round_checks.sh r04_final2 prof
cd "$R"
timeout 400 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-200 "$OUT/bench_cfg4.json" | tail -1
timeout 700 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1 > "$OUT/bench_cfg5_b256.json" 2> "$OUT/bench_cfg5_b256.err"; cut -c1-200 "$OUT/bench_cfg5_b256.json" | tail -1
