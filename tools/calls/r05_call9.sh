#!/bin/bash
# round 5, call 9: sweep of the weight-gradient workgroup target (SIMCLR_WGRAD_BLOCKS, default 1536) in the bf16 step and in the parity mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call9
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
for rep in 1 2; do
for w in 1536 768 1024 2048 3072; do
  SIMCLR_WGRAD_BLOCKS=$w timeout 200 $B > "$OUT/bf16_${w}_$rep.json" 2>> "$OUT/err.txt"
done
done
for w in 1536 768 1024 2048 3072; do
  SIMCLR_WGRAD_BLOCKS=$w timeout 200 $B --dtype f32 --f32_matmul bf16x6_3 --steps 6 --warmup 2 > "$OUT/parity_${w}_1.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], 'wgrad', d['kernels']['conv_wgrad']['ms_per_step'])
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
