#!/bin/bash
# round 6, call 20: stem weight gradient from pre-split operands (stem_wgrad_ps) -- tests, A/B inside the step, kernel times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call20
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "stem" > "$OUT/pytest_stem.txt" 2>&1; tail -15 "$OUT/pytest_stem.txt"
SIMCLR_STEM_WGRAD_PS_STAGES=3 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "stem_conv" > "$OUT/pytest_stem3.txt" 2>&1; tail -3 "$OUT/pytest_stem3.txt"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "train_step_resnet50_224_batch32 or reference_source_fixtures" > "$OUT/pytest_step.txt" 2>&1; tail -3 "$OUT/pytest_step.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_STEM_WGRAD_PS=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  env SIMCLR_STEM_WGRAD_PS=1 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
  env SIMCLR_STEM_WGRAD_PS=1 SIMCLR_STEM_WGRAD_PS_STAGES=3 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new3_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_wgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
cd /tmp; export TMPDIR=/tmp
mkdir -p "$OUT/new"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/new" -o stats -- $B --no_kernel_events --steps 3 --warmup 1 > "$OUT/new/prof.log" 2>&1
rm -f "$OUT"/*/*kernel_trace.csv "$OUT"/*/*agent_info.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/new/stats_kernel_stats.csv')))
for r in rows:
    n = r['Name']
    if any(w in n for w in ('stem', 'pool', 'presplit_packed', 'pack_views', 'bn_bwd_apply<float, 2, true')):
        print('  %8.1f us x %5.1f  %s' % (float(r['AverageNs']) / 1e3, int(r['Calls']) / 4, n[:110]))
PY
cd "$R"; tail -3 "$OUT/err.txt"; echo "total: $((SECONDS - T0)) s"
