#!/bin/bash
# round 6, call 57: stem BatchNorm-backward apply with a thread per 2 x 2 block of input pixels (SIMCLR_POOL_APPLY_2X2=0 = per pixel) -- tests, A/B, kernel time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call57
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "maxpool or pool or stem or batch32 or resnet18 or reference_source_fixtures or determinis" > "$OUT/pytest.txt" 2>&1; tail -3 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_POOL_APPLY_2X2=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o s -- $B --no_kernel_events --steps 2 --warmup 1 > "$OUT/prof.log" 2>&1
grep -i "reduce_pool\|apply_pool" "$OUT"/prof/*kernel_stats.csv | cut -c1-160
rm -f "$OUT"/prof/*trace.csv "$OUT"/prof/*agent_info.csv
