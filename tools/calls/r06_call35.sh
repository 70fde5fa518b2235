#!/bin/bash
# round 6, call 35: fp32 halo-window path of the 3x3 stride-1 forward / data gradient -- tests, A/B (SIMCLR_CONV3_WIN32=0), per-layer table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call35
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "presplit_gradient or split_bf16 or bench_path or presplit_weight or batch32 or reference_source_fixtures or parity_at_baseline or resnet18 or sk_ or conv_fwd_dgrad" > "$OUT/pytest.txt" 2>&1; tail -6 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_CONV3_WIN32=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_igemm_fwd', {}).get('ms_per_step'), d['kernels'].get('conv_igemm_dgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
cd "$R"
SIMCLR_CONV3_WIN32=0 timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps > "$OUT/per_layer_gather.txt" 2>&1
timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps > "$OUT/per_layer_window.txt" 2>&1
grep "k3 s1" "$OUT/per_layer_gather.txt"; grep "k3 s1" "$OUT/per_layer_window.txt"
