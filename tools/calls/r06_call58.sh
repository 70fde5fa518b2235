#!/bin/bash
# round 6, call 58: FINAL build (call 56 + the 2 x 2-block stem apply): suite, smoke, default bench line, rocprofv3 stats + trace, PMC passes, per-layer table, cfg4 / cfg5
# bf16 speed mode; the two PMC passes; per-layer table of the headline mode; cfg4 / cfg5 lines in the headline mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call58
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
cd "$R"
timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_parity.txt" 2>&1; tail -3 "$OUT/per_layer_parity.txt"
BC="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
timeout 600 $BC --width_multiplier 2 --sk_ratio 0.0625 --per_gpu_batch 512 > "$OUT/cfg4_b512_parity.json" 2>> "$OUT/err.txt"
timeout 900 $BC --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 128 > "$OUT/cfg5_b128_parity.json" 2>> "$OUT/err.txt"
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/cfg*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], 'mfma_frac', d['step_mfma_frac'], 'peak_hbm_gb', d['peak_hbm_gb'])
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
ls "$OUT"; echo "total: $((SECONDS - T0)) s"
