#!/bin/bash
# round 6, call 2: pre-split (hi, lo) bf16 gradient between BatchNorm backward and the convolution backward GEMMs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call2
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "presplit_gradient or probes" > "$OUT/pytest_ps.txt" 2>&1
tail -15 "$OUT/pytest_ps.txt"
timeout 900 python tools/step_modes.py --modes f16x3_3 --out "$OUT/step_modes.json" > "$OUT/step_modes.txt" 2>&1
tail -14 "$OUT/step_modes.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2 --dtype f32 --f32_matmul f16x3_3"
for rep in 1 2; do
  SIMCLR_PS_BWD=0 timeout 300 $B > "$OUT/bench_ps0_$rep.json" 2>> "$OUT/err.txt"
  SIMCLR_PS_BWD=1 timeout 300 $B > "$OUT/bench_ps1_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if v.get('ms_per_step', 0) > 1.0})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -5 "$OUT/err.txt"
