#!/bin/bash
# round 6, call 22: full GPU suite on the build with the stem work (rolling forward, pre-split weight gradient, fused pool backward by default in
# fp32, one workgroup per channel in conv_pivot_row); three default-mode step timings + kernel stats
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call22
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > "$OUT/pytest_gpu.txt" 2>&1; tail -12 "$OUT/pytest_gpu.txt"; echo "suite: $((SECONDS - T0)) s"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_$rep.json" 2>> "$OUT/err.txt"
done
timeout 300 $B --dtype bf16 --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_bf16_unfused.json" 2>> "$OUT/err.txt"
SIMCLR_POOL_FUSION=1 timeout 300 $B --dtype bf16 --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_bf16_fused.json" 2>> "$OUT/err.txt"
timeout 300 $B --dtype bf16 --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_bf16_unfused2.json" 2>> "$OUT/err.txt"
SIMCLR_POOL_FUSION=1 timeout 300 $B --dtype bf16 --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_bf16_fused2.json" 2>> "$OUT/err.txt"
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'])
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
cd /tmp; export TMPDIR=/tmp
mkdir -p "$OUT/prof"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- $B --no_kernel_events --steps 3 --warmup 1 > "$OUT/prof/prof.log" 2>&1
gzip -f "$OUT"/prof/*kernel_trace.csv; rm -f "$OUT"/*/*agent_info.csv
cd "$R"; tail -3 "$OUT/err.txt"; echo "total: $((SECONDS - T0)) s"
