#!/bin/bash
# round 6, call 48: experimental 256 x 256 / eight-waves-along-k weight-gradient tile for the fp32 1x1 layers (SIMCLR_WGRAD_F32_256=1) -- test, per layer, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call48
mkdir -p "$OUT"
cd "$R"
SIMCLR_WGRAD_F32_256=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "presplit_gradient" > "$OUT/pytest.txt" 2>&1; tail -3 "$OUT/pytest.txt"
for r in 0 1; do
  SIMCLR_WGRAD_F32_256=$r timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_$r.txt" 2>&1
  echo "== 256-tile $r"; grep "k1 s1" "$OUT/per_layer_$r.txt" | awk '{print $1, $2, $3, $4, $9}'
done
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  env SIMCLR_WGRAD_F32_256=1 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
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
