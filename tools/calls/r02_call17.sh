#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call17
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_on_$v.json" 2> "$OUT/bench.err"
SIMCLR_PREP_BATCH=0 timeout 200 $B > "$OUT/bench_off_$v.json" 2> "$OUT/bench_off.err"
done
for f in on_a off_a on_b off_b; do
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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o st -- python $R/bench.py --steps 4 --warmup 4 --no_cpu_baseline --no_f32 --prof_steps 0 > "$OUT/prof.log" 2>&1
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof"
head -48 "$OUT/kernel_stats.csv" | cut -c1-150
