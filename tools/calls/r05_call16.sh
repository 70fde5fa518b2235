#!/bin/bash
# round 5, call 16: profile set of the final defaults: bench line (default flags), rocprofv3 kernel stats + trace and the two PMC passes of the
# bf16 step, kernel stats of the parity mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call16
mkdir -p "$OUT/parity"
cd "$R"
T0=$SECONDS
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-200 "$OUT/bench.json" | tail -1; echo "bench: $((SECONDS - T0)) s"
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc --no_parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/parity" -o stats -- $B --dtype f32 --f32_matmul bf16x6_3 --steps 3 --warmup 1 > "$OUT/parity/prof.log" 2>&1
gzip -f "$OUT"/parity/*kernel_trace.csv 2>/dev/null
rm -f "$OUT"/*agent_info.csv "$OUT"/parity/*agent_info.csv
cd "$R"; ls "$OUT"; echo "total: $((SECONDS - T0)) s"
