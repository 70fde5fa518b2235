#!/bin/bash
# round 6, call 21: fused stem backward (max-pool backward inside the BatchNorm backward) with the clamped four-window gather and a pre-split
# output for the stem's weight gradient, against the unfused kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call21
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pool or stem" > "$OUT/pytest_pool.txt" 2>&1; tail -5 "$OUT/pytest_pool.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_POOL_FUSION=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_unfused_$rep.json" 2>> "$OUT/err.txt"
  env SIMCLR_POOL_FUSION=1 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_fused_$rep.json" 2>> "$OUT/err.txt"
done
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
mkdir -p "$OUT/fused"
env SIMCLR_POOL_FUSION=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fused" -o stats -- $B --no_kernel_events --steps 3 --warmup 1 > "$OUT/fused/prof.log" 2>&1
rm -f "$OUT"/*/*kernel_trace.csv "$OUT"/*/*agent_info.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/fused/stats_kernel_stats.csv')))
for r in rows:
    n = r['Name']
    if any(w in n for w in ('stem', 'pool', 'presplit_packed')):
        print('  %8.1f us x %5.1f  %s' % (float(r['AverageNs']) / 1e3, int(r['Calls']) / 4, n[:110]))
PY
cd "$R"; tail -3 "$OUT/err.txt"; echo "total: $((SECONDS - T0)) s"
