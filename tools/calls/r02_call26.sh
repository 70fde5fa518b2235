#!/bin/bash
# mini refresh of the judged measurements for the final build: bench line, blur-on run, rocprofv3 kernel stats, two PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_final2
mkdir -p "$OUT"
cd "$R"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-260 "$OUT/bench.json" | tail -1
timeout 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --prof_steps 0 --use_blur > "$OUT/bench_blur.json" 2> "$OUT/bench_blur.err"; cut -c1-260 "$OUT/bench_blur.json" | tail -1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
rm -f "$OUT"/*kernel_trace.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
ls "$OUT"
