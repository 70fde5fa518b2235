#!/bin/bash
# round 4, call 9 (re-entry; call 8's outputs were lost with the container): parity subset for the wide-tile fp32-accumulator
# statistics, then smoke, the default bench line, rocprofv3 kernel stats / trace and the two PMC passes on the committed defaults
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call9
mkdir -p "$OUT"
cd "$R"
timeout 700 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wide or gram or bitwise or 256_tile or determinis or test_train_step_bf16 or fixed_thresholds or bench_path_shapes" > "$OUT/pytest_subset.log" 2>&1
tail -3 "$OUT/pytest_subset.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_subset.log" | head -20 | cut -c1-250
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json" | tail -1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
ls "$OUT"
