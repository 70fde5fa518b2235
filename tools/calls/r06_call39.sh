#!/bin/bash
# round 6, call 39: weight gradient on a plain fp32 gradient: the tile split in LDS once per chunk (PSD = 2), four waves along k -- tests, A/B (SIMCLR_WGRAD_LDSPS=0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call39
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "split_bf16 or wgrad or fold or presplit_gradient or batch32 or reference_source_fixtures or parity_at_baseline or resnet18 or sk_ or conv_fwd_dgrad" > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_WGRAD_LDSPS=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
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
tail -3 "$OUT/err.txt"
