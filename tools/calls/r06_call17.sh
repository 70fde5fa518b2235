#!/bin/bash
# round 6, call 17: the stem in the parity mode -- rolling-fragment forward (stem_conv_fwd<float, ., 14, 13>), three-deep ring for the stem weight
# gradient, max-pool backward fused into the stem BatchNorm backward (opt-in since round 2; fp32 storage doubles the bytes it removes);
# two-replica default-mode test with the corrected gates
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call17
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem" > "$OUT/pytest_stem.txt" 2>&1; tail -3 "$OUT/pytest_stem.txt"
SIMCLR_STEM_WGRAD_STAGES=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem_conv" > "$OUT/pytest_stem_s3.txt" 2>&1; tail -3 "$OUT/pytest_stem_s3.txt"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "fast_parity_mode" > "$OUT/pytest_dist.txt" 2>&1; tail -4 "$OUT/pytest_dist.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
OLD="SIMCLR_STEM_ROLL=0 SIMCLR_STEM_WGRAD_STAGES=2 SIMCLR_POOL_FUSION=0"
NEW="SIMCLR_STEM_ROLL=1 SIMCLR_STEM_WGRAD_STAGES=3 SIMCLR_POOL_FUSION=1"
for rep in 1 2 3; do
  env $OLD timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  env $NEW timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
env SIMCLR_STEM_ROLL=1 SIMCLR_STEM_WGRAD_STAGES=2 SIMCLR_POOL_FUSION=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_rollonly_1.json" 2>> "$OUT/err.txt"
env SIMCLR_STEM_ROLL=1 SIMCLR_STEM_WGRAD_STAGES=3 SIMCLR_POOL_FUSION=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_rollstages_1.json" 2>> "$OUT/err.txt"
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if v.get('ms_per_step', 0) > 0.5})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
cd /tmp; export TMPDIR=/tmp
mkdir -p "$OUT/old" "$OUT/new"
env $OLD timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/old" -o stats -- $B --no_kernel_events --steps 3 --warmup 1 > "$OUT/old/prof.log" 2>&1
env $NEW timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/new" -o stats -- $B --no_kernel_events --steps 3 --warmup 1 > "$OUT/new/prof.log" 2>&1
rm -f "$OUT"/*/*kernel_trace.csv "$OUT"/*/*agent_info.csv
python - <<PY
import csv
for tag in ('old', 'new'):
    rows = list(csv.DictReader(open('$OUT/%s/stats_kernel_stats.csv' % tag)))
    print(tag, 'total ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 4e6)
    for r in rows:
        n = r['Name']
        if any(w in n for w in ('stem', 'pool', '256, 64, 2,', 'bn_bwd_apply<float, 2, true, false>', 'bn_bwd_reduce')):
            print('  %8.1f us x %5.1f  %s' % (float(r['AverageNs']) / 1e3, int(r['Calls']) / 4, n[:110]))
PY
cd "$R"; tail -3 "$OUT/err.txt"; echo "total: $((SECONDS - T0)) s"
