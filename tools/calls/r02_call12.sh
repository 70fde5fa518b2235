#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call12
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
for w in 1 0; do
SIMCLR_CONV3_WIN=$w timeout 300 python tools/microbench.py --what conv --out "$OUT/micro_win$w.json" > "$OUT/micro_win$w.log" 2>&1
grep " 3 \| 3x3\|k3" "$OUT/micro_win$w.log" | head; 
done
grep -n "x3\b" "$OUT/micro_win1.log" | head -3
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
SIMCLR_CONV3_WIN=1 timeout 200 $B > "$OUT/bench_on_$v.json" 2> "$OUT/bench_on.err"
SIMCLR_CONV3_WIN=0 timeout 200 $B > "$OUT/bench_off_$v.json" 2> "$OUT/bench_off.err"
done
for f in on_a off_a on_b off_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], d.get('augment'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench_on.err"
