#!/bin/bash
# round 6, call 44: nine-tap weight gradient in fp32 storage (conv_wgrad3x3_f32ps: window split in LDS, pre-split gradient) -- tests, per-layer, A/B (SIMCLR_WGRAD_3X3=0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call44
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "presplit_gradient or batch32 or reference_source_fixtures or parity_at_baseline or resnet18 or sk_ or wgrad" > "$OUT/pytest.txt" 2>&1; tail -6 "$OUT/pytest.txt"
SIMCLR_WGRAD_3X3=0 timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_pertap.txt" 2>&1
timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_ninetap.txt" 2>&1
grep "k3 s1" "$OUT/per_layer_pertap.txt"; grep "k3 s1" "$OUT/per_layer_ninetap.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_WGRAD_3X3=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
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
