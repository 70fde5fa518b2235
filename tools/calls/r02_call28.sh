#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call28
mkdir -p "$OUT"
cd "$R"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "deterministic or fused_conv3 or resnet18_f32 or checkpoint" > "$OUT/pytest.log" 2>&1
tail -2 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --prof_steps 0"
for v in a b c; do
for L in 1 2; do
SIMCLR_CONV3_FUSED=$L timeout 200 $B > "$OUT/bench_${L}_$v.json" 2> "$OUT/bench.err"
done
done
for f in 1_a 2_a 1_b 2_b 1_c 2_c; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms']['median'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
