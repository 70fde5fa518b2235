#!/bin/bash
# round 6, call 25: packed views and their pre-split copy written in one pass (simclr_pack_views_ps)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call25
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem or batch32_fast_parity or reference_source_fixtures" > "$OUT/pytest.txt" 2>&1; tail -3 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d['ms_per_step'])
PY
tail -3 "$OUT/err.txt"
