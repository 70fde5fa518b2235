#!/bin/bash
# round 5, call 10: tests around the weight gradient after the workgroup-target change, a sweep of the nine-tap kernel's target, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call10
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad or bench_path or wgrad or deterministic or fold or gram or train_step_bf16 or fixed_thresholds" > "$OUT/pytest.log" 2>&1; tail -2 "$OUT/pytest.log" | cut -c1-200
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
for rep in 1 2; do
for w in default 384 768 1024; do
  if [ $w = default ]; then timeout 200 $B > "$OUT/w3_${w}_$rep.json" 2>> "$OUT/err.txt"; else SIMCLR_WGRAD3_BLOCKS=$w timeout 200 $B > "$OUT/w3_${w}_$rep.json" 2>> "$OUT/err.txt"; fi
done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/w3_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], 'wgrad', d['kernels']['conv_wgrad']['ms_per_step'])
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
