#!/bin/bash
# round 6, call 28: strided projection shortcuts without zero fill (sparse store + parity accumulate) -- tests, A/B (SIMCLR_SPARSE_DGRAD=0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call28
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "zero_fill or dgrad or batch32 or reference_source_fixtures or sk_ or parity_at_baseline or resnet18 or r18" > "$OUT/pytest.txt" 2>&1; tail -6 "$OUT/pytest.txt"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "fast_parity_mode or equals_global" > "$OUT/pytest_dist.txt" 2>&1; tail -3 "$OUT/pytest_dist.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_SPARSE_DGRAD=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_igemm_dgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
