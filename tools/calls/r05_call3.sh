#!/bin/bash
# round 5, call 3: the profile set of the round -- bench line (default K / W, CPU baseline, both fp32 modes, measured parity, PMC sampled in-run),
# rocprofv3 kernel stats + trace and the two PMC passes for the bf16 headline AND for the parity mode, per-layer microbench in both modes,
# cfg4, the CPU thread sweeps (run beside the PMC passes, whose timings do not matter)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call3
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json" | tail -1; echo "bench: $((SECONDS - T0)) s"
timeout 300 python tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.txt" 2>&1; tail -2 "$OUT/microbench.txt"
timeout 300 python tools/microbench.py --dtype f32 --f32_matmul bf16x6_3 --what conv --iters 3 --out "$OUT/microbench_parity.json" > "$OUT/microbench_parity.txt" 2>&1; tail -2 "$OUT/microbench_parity.txt"
timeout 400 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --no_parity --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-260 "$OUT/bench_cfg4.json" | tail -1
echo "timed part done: $((SECONDS - T0)) s"
# ---- from here on timings do not matter: CPU thread probes in the background
( timeout 400 python tools/oracle_threads_probe.py --out "$OUT/oracle_threads.json" > "$OUT/oracle_threads.txt" 2>&1;
  timeout 300 python tools/cpu_threads_sweep.py --out "$OUT/cpu_threads.json" > "$OUT/cpu_threads.txt" 2>&1 ) &
PROBE=$!
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc --no_parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
P="$B --dtype f32 --f32_matmul bf16x6_3"
mkdir -p "$OUT/parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/parity" -o stats -- $P --steps 3 --warmup 1 > "$OUT/parity/prof.log" 2>&1
gzip -f "$OUT"/parity/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/parity/pmc_f" -o f -- $P --steps 1 --warmup 1 > "$OUT/parity/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/parity/pmc_w" -o w -- $P --steps 1 --warmup 1 > "$OUT/parity/pmc_w.log" 2>&1
gzip -f "$OUT"/parity/pmc_f/*counter_collection.csv "$OUT"/parity/pmc_w/*counter_collection.csv 2>/dev/null
NT="python $R/tools/microbench.py --what ntxent --iters 3 --out $OUT/mb_ntxent.json"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_nt_f" -o f -- $NT > "$OUT/pmc_nt_f.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_nt_w" -o w -- $NT > "$OUT/pmc_nt_w.log" 2>&1
wait $PROBE
cd "$R"
cat "$OUT/oracle_threads.txt" | tail -14; cat "$OUT/cpu_threads.txt" | tail -5
rm -f "$OUT"/*agent_info.csv "$OUT"/parity/*agent_info.csv
du -sh "$OUT"; ls "$OUT" "$OUT/parity"; echo "total: $((SECONDS - T0)) s"
