#!/bin/bash
# round 6, call 30: fp32 LDS bank key of the weight-gradient / Gram tiles (px_key<float>) -- tests, A/B against the HEAD build (libsimclr_hip_a.so), conflict counters
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call30
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "wgrad or gram or conv_fwd_dgrad or bench_path or fold or batch32 or reference_source_fixtures or split_bf16" > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_a.so timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_wgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
cd /tmp; export TMPDIR=/tmp
B2="$B --no_kernel_events"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc" -o l -- $B2 --steps 1 --warmup 1 > "$OUT/pmc.log" 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/pmc/*counter_collection.csv')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:120]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    rows = sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:25]
    for n, c in rows:
        a = c.get('SQ_LDS_IDX_ACTIVE', 0); b = c.get('SQ_LDS_BANK_CONFLICT', 0)
        print('%6.3f conflict/active  active %.3e  %s' % (b / a if a else 0, a, n))
PY
gzip -f "$OUT"/pmc/*counter_collection.csv 2>/dev/null; rm -f "$OUT"/pmc/*agent_info.csv "$OUT"/pmc/*kernel_trace.csv
