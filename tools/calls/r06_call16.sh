#!/bin/bash
# round 6, call 16 (= call 13 re-run on the final build): full GPU suite + smoke on the final defaults; profile set of the headline (parity) mode and of the bf16 speed mode:
# default bench line, rocprofv3 kernel stats + trace, the two PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call16
mkdir -p "$OUT/speed"
cd "$R"
T0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > "$OUT/pytest_gpu.txt" 2>&1; tail -14 "$OUT/pytest_gpu.txt"; echo "suite: $((SECONDS - T0)) s"
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -3 "$OUT/smoke.txt"
T1=$SECONDS
timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json" | tail -1; echo "bench: $((SECONDS - T1)) s"
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc --no_parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/speed" -o stats -- $B --dtype bf16 --steps 3 --warmup 1 > "$OUT/speed/prof.log" 2>&1
gzip -f "$OUT"/speed/*kernel_trace.csv 2>/dev/null
rm -f "$OUT"/*agent_info.csv "$OUT"/speed/*agent_info.csv
cd "$R"; ls "$OUT"; echo "total: $((SECONDS - T0)) s"
