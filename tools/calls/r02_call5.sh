#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call5
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "stem or ntxent or 2x_sk or pooling or model_api" > "$OUT/pytest.log" 2>&1
tail -6 "$OUT/pytest.log"
timeout 600 python tools/bf16_parity_report.py structured > "$OUT/bf16_report.log" 2>&1; grep -v "grad-error" "$OUT/bf16_report.log" | tail -20; grep "grad-error" "$OUT/bf16_report.log" | tail -8 | cut -c1-160
cp gpurun_out/bf16_parity.json "$OUT/" 2>/dev/null
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
SIMCLR_STEM_WGRAD_MT=0 timeout 200 $B > "$OUT/bench_nostemmt.json" 2> "$OUT/bench_nostemmt.err"
for f in base nostemmt; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], d['ntxent'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
done
