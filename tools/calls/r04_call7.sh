#!/bin/bash
# round 4, call 7: NT-Xent forward / backward as ONE launch each (last-arriver reductions): parity + timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call7
mkdir -p "$OUT"
cd "$R"
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ntxent or resnet18_f32 or test_train_step_bf16 or determinis or free_proj" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -20 | cut -c1-250
timeout 200 python tools/microbench.py --what ntxent --iters 9 --out "$OUT/ntxent.json" > "$OUT/ntxent.log" 2>&1
tail -6 "$OUT/ntxent.log"
timeout 200 python tools/stress_race.py > "$OUT/stress.log" 2>&1; tail -3 "$OUT/stress.log"
timeout 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --no_pmc > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<'EOP'
import json,os
f=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call7/bench.json'
try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print('%8.3f ms %8.1f img/s' % (d['ms_per_step'], d['value'])); print(d['ntxent'])
except Exception as e: print('ERR', e, open(f.replace('.json','.err')).read()[-800:])
EOP
