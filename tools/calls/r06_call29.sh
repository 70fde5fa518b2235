#!/bin/bash
# round 6, call 29: LDS bank-conflict counters of the parity-mode kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call29
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "lds|LDS" | head -40 > "$OUT/avail_lds.txt"; head -30 "$OUT/avail_lds.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc" -o l -- $B --steps 1 --warmup 1 > "$OUT/pmc.log" 2>&1
tail -3 "$OUT/pmc.log"
ls "$OUT/pmc"
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/pmc/*counter_collection.csv')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:120]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    rows = sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:25]
    for n, c in rows:
        a = c.get('SQ_LDS_IDX_ACTIVE', 0); b = c.get('SQ_LDS_BANK_CONFLICT', 0)
        print('%6.3f conflict/active  active %.3e  insts %.3e  %s' % (b / a if a else 0, a, c.get('SQ_INSTS_LDS', 0), n))
PY
gzip -f "$OUT"/pmc/*counter_collection.csv 2>/dev/null; rm -f "$OUT"/pmc/*agent_info.csv "$OUT"/pmc/*kernel_trace.csv
