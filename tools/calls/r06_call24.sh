#!/bin/bash
# round 6, call 24: the 64-wide rule for the fp32 data-gradient launches only (default K <= 128) against 128-wide tiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call24
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
SIMCLR_IGEMM_BN64_K32=128 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split_bf16 or tail_f32 or dgrad_with_fused or presplit or batch32_fast_parity" > "$OUT/pytest_k128.txt" 2>&1; tail -4 "$OUT/pytest_k128.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  for k in 0 128; do
    env SIMCLR_IGEMM_BN64_K32=$k timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_k${k}_$rep.json" 2>> "$OUT/err.txt"
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in d['kernels'].items() if v.get('ms_per_step', 0) > 5})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
cd "$R"; tail -3 "$OUT/err.txt"; echo "total: $((SECONDS - T0)) s"
