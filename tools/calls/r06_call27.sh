#!/bin/bash
# round 6, call 27: which pre-split launches are left in the step (kernel stats with and without SIMCLR_PS_W)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call27
mkdir -p "$OUT/new" "$OUT/old"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "presplit_weight" > "$OUT/pytest.txt" 2>&1; tail -2 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/new" -o stats -- $B --steps 3 --warmup 1 > "$OUT/new/prof.log" 2>&1
SIMCLR_PS_W=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/old" -o stats -- $B --steps 3 --warmup 1 > "$OUT/old/prof.log" 2>&1
rm -f "$OUT"/*/*kernel_trace.csv "$OUT"/*/*agent_info.csv
python - <<PY
import csv
for tag in ('old', 'new'):
    rows = list(csv.DictReader(open('$OUT/%s/stats_kernel_stats.csv' % tag)))
    print(tag, 'total ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 4e6)
    for r in rows:
        n = r['Name']
        if any(w in n for w in ('presplit', 'prep_weights', 'pivot')):
            print('  %8.1f us x %6.1f  %s' % (float(r['AverageNs']) / 1e3, int(r['Calls']) / 4, n[:100]))
PY
