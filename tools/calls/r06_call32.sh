#!/bin/bash
# round 6, call 32: fp16 split with one mixed-precision fma per element (v_fma_mixlo/mixhi_f16) -- piece test against torch's rounding, forward tests,
# A/B against the previous build (libsimclr_hip_a.so), LDS conflict counters of the stem kernels at the new pitch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call32
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "presplit or pieces or split_bf16 or bench_path or stem or fused_bn_apply or batch32 or reference_source_fixtures or ntxent or parity_at_baseline or ps_" > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
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
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_igemm_fwd', {}).get('ms_per_step'), d['kernels'].get('stem_conv_fwd', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc" -o l -- $B --no_kernel_events --steps 1 --warmup 1 > "$OUT/pmc.log" 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/pmc/*counter_collection.csv')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:120]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:40]:
        a = c.get('SQ_LDS_IDX_ACTIVE', 0); b = c.get('SQ_LDS_BANK_CONFLICT', 0)
        if b > 0: print('%6.3f conflict/active  active %.3e  %s' % (b / a if a else 0, a, n))
PY
rm -rf "$OUT"/pmc
