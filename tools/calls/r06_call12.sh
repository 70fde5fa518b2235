#!/bin/bash
# round 6, call 12: split tail for the fp32 three-term launches -- gates + A/B in the step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call12
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split_bf16_bench_path or presplit_gradient or fast_parity_mode or split_bf16_matmul or baseline_sizes" > "$OUT/pytest_sel.txt" 2>&1; tail -5 "$OUT/pytest_sel.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
for rep in 1 2 3; do
  SIMCLR_IGEMM_SPLIT_F32=0 timeout 300 $B > "$OUT/bench_nosplit_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B > "$OUT/bench_split_$rep.json" 2>> "$OUT/err.txt"
done
SIMCLR_IGEMM_SPLIT_MINSTEPS=4 timeout 300 $B > "$OUT/bench_split_min4.json" 2>> "$OUT/err.txt"
SIMCLR_IGEMM_SPLIT_MINSTEPS=16 timeout 300 $B > "$OUT/bench_split_min16.json" 2>> "$OUT/err.txt"
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
tail -3 "$OUT/err.txt"
