#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call7
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "folded" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_fold.json" 2> "$OUT/bench_fold.err"
python - "$OUT/bench_fold.json" fold <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
timeout 400 python tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.txt" 2>&1; tail -45 "$OUT/microbench.txt"
