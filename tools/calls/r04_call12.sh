#!/bin/bash
# round 4, call 12: the profile set of the final build (bench line, rocprofv3 kernel stats / trace, two PMC passes), the new kernel tests,
# smoke, per-layer microbench, cfg4 line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final4
mkdir -p "$OUT"
cd "$R"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json" | tail -1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
cd "$R"
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "pivoted or presplit or split_bf16_matmul" > "$OUT/pytest_new.log" 2>&1; tail -2 "$OUT/pytest_new.log" | cut -c1-200; grep -n "^FAILED\|^E  " "$OUT/pytest_new.log" | head -10 | cut -c1-250
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 200 python tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.txt" 2>&1; tail -2 "$OUT/microbench.txt" | cut -c1-200
timeout 150 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-200 "$OUT/bench_cfg4.json" | tail -1
ls "$OUT"
