#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call16
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "not fixed_thresholds and not iid" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_$v.json" 2> "$OUT/bench.err"
done
for f in a b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o st -- python $R/bench.py --steps 4 --warmup 4 --no_cpu_baseline --no_f32 --prof_steps 0 > "$OUT/prof.log" 2>&1
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/prof" -name "*.db" -delete 2>/dev/null; find "$OUT/prof" -name "*kernel_trace.csv" -delete 2>/dev/null
head -45 "$OUT/kernel_stats.csv" | cut -c1-150
