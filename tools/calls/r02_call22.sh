#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call22
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --prof_steps 0"
for v in a b; do
for L in 0 1 2; do
SIMCLR_CONV3_FUSED=$L timeout 200 $B > "$OUT/bench_${L}_$v.json" 2> "$OUT/bench.err"
done
done
for f in 0_a 1_a 2_a 0_b 1_b 2_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], d['train_metrics']['train/total_loss'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
cd /tmp && export TMPDIR=/tmp
SIMCLR_CONV3_FUSED=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o st -- python $R/bench.py --steps 4 --warmup 4 --no_cpu_baseline --no_f32 --prof_steps 0 > "$OUT/prof.log" 2>&1
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/prof" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_trace.csv"
rm -rf "$OUT/prof"
python - "$OUT/kernel_trace.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step's FAPPLY / stats-only kernels with durations and grid
n=len(rows)
sel=[r for r in rows[int(n*0.86):] if 'conv_igemm_persistent' in r['Kernel_Name'] and ('false, false, false, false, true' in r['Kernel_Name'] or '0, 128, 128, 4, 2, true, false, false, false, false' in r['Kernel_Name'] or '0, 128, 64, 4, 2, true, false, false, false, false' in r['Kernel_Name'])]
for r in sel[:80]:
    print((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r['Kernel_Name'][40:110])
PY
gzip -f "$OUT/kernel_trace.csv"
