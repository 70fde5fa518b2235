#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call6
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "folded or resnet50_f32_reference_init or fixed_thresholds or deterministic" > "$OUT/pytest.log" 2>&1
tail -12 "$OUT/pytest.log" | cut -c1-250; grep -n "bn_fold\|Error" "$OUT/pytest.log" | head -30 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_fold.json" 2> "$OUT/bench_fold.err"
SIMCLR_BN_FOLD=0 timeout 200 $B > "$OUT/bench_nofold.json" 2> "$OUT/bench_nofold.err"
for f in fold nofold; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}, d['train_metrics'])
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
done
