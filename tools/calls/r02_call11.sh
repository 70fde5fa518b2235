#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call11
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "folded or supervised_head or resnet50_f32_reference_init or deterministic or resnet18_f32" > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_on_$v.json" 2> "$OUT/bench_on.err"
done
for f in on_a on_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['traffic'], d['roofline'].get('traffic_over_algorithmic'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench_on.err"
