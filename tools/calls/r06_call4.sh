#!/bin/bash
# round 6, call 4: full GPU suite on the round-6 tree (split-fp16 forward, pre-split gradients, per-call terms, SGD / Adam kernels,
# baseline-size parity tests) + the default bench line (headline = tolerance-meeting mode)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call4
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > "$OUT/pytest_gpu.txt" 2>&1
tail -45 "$OUT/pytest_gpu.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_err.txt"
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
    print('value', d['value'], d['unit'], 'ms', d['ms_per_step'], 'dtype', d['dtype'], d.get('f32_matmul'), 'north_star_met', d.get('north_star_met'))
    print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'hbm_frac', 'mfma_frac', 'ms_per_step', 'step_total_traffic_gb')})
    print('families', d.get('families'))
    for k in ('speed_mode', 'f32_mode', 'parity_mode_bf16x6'):
        m = d.get(k) or {}
        print(k, m.get('value'), m.get('ms_per_step'), (m.get('roofline') or {}).get('frac'))
    print('parity', {k: (v['loss_rel'], v['emb_abs'], v['north_star_met']) for k, v in d['parity']['modes'].items()})
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 "$OUT/bench_err.txt"
