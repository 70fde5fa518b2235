#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call14
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv or folded or deterministic or resnet18_f32" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
for w in 1 0; do
SIMCLR_EPI_PREFETCH=$w timeout 300 python tools/microbench.py --what conv --out "$OUT/micro_pf$w.json" > "$OUT/micro_pf$w.log" 2>&1
echo "== prefetch $w"; grep "k3 s1\|56x56 64\|28x28 128->512\|14x14 256->1024\|totals" "$OUT/micro_pf$w.log"
done
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
SIMCLR_EPI_PREFETCH=1 timeout 200 $B > "$OUT/bench_pf1_$v.json" 2> "$OUT/bench_pf1.err"
SIMCLR_EPI_PREFETCH=0 timeout 200 $B > "$OUT/bench_pf0_$v.json" 2> "$OUT/bench_pf0.err"
done
for f in pf1_a pf0_a pf1_b pf0_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench_pf1.err"
