#!/bin/bash
# round 5, call 1: the product against the reference-source fixtures on the device (model + single_step, exact and bf16x6_3), the
# split-tail replay test, the peer-mapped exchange soak, and a first bench line with the measured parity block + parity-mode roofline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call1
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --durations=30 -k "reference_source_fixtures or hipgraph or split_tail" > "$OUT/pytest_pin.log" 2>&1
tail -45 "$OUT/pytest_pin.log" | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q --durations=10 -k "equals_global_batch_oracle or peer_mapped" > "$OUT/pytest_dist.log" 2>&1
tail -25 "$OUT/pytest_dist.log" | cut -c1-400
timeout 500 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_pmc > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -5 "$OUT/bench.err" | cut -c1-300
python - <<PY
import json
d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('parity', json.dumps(d['parity'])[:1800])
print('ntxent', d['ntxent'])
pm = d['parity_mode']
print('parity_mode', {k: v for k, v in pm.items() if k not in ('families', 'other_kernels')})
print('families', pm.get('families'))
print('f32_mode', d['f32_mode'])
PY
