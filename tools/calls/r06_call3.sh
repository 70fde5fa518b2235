#!/bin/bash
# round 6, call 3: pre-split gradient kernel tests after the contraction fix + rocprofv3 kernel stats of the f16x3_3 parity step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call3
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "presplit_gradient" > "$OUT/pytest_ps.txt" 2>&1
tail -8 "$OUT/pytest_ps.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o parity --output-format csv -- python "$R/bench.py" --steps 4 --warmup 2 --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events > "$OUT/bench_prof.json" 2>> "$OUT/err.txt"
cd "$R"
python - <<PY
import csv, glob
fs = glob.glob('$OUT/prof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over 6 steps', tot / 1e6)
for r in rows[:40]:
    print('%-150s %5s %9.3f ms/step %7.1f us' % (r['Name'][:150], r['Calls'], float(r['TotalDurationNs']) / 1e6 / 6, float(r['AverageNs']) / 1e3))
PY
tail -3 "$OUT/err.txt"
