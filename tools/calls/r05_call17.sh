#!/bin/bash
# round 5, call 17: second interleaved sweep around the new defaults
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call17
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/$name.json" 2>> "$OUT/err.txt"; }
for rep in 1 2; do
  run default_$rep A=1
  run bn64k32_$rep SIMCLR_IGEMM_BN64_K=32
  run cls2_$rep SIMCLR_IGEMM_256_CLASSES=2
  run cls4_$rep SIMCLR_IGEMM_256_CLASSES=4
  run wg896_$rep SIMCLR_WGRAD_BLOCKS=896
  run wg1280_$rep SIMCLR_WGRAD_BLOCKS=1280
  run c3f1_$rep SIMCLR_CONV3_FUSED=1
  run wide0_$rep SIMCLR_IGEMM_WIDE=0
  run default_b_$rep A=1
done
python - <<PY
import json, glob, os
rows = {}
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = os.path.basename(f)[:-5].rsplit('_', 1)[0]
        rows.setdefault(k, []).append(d['ms_per_step'])
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
for k, v in sorted(rows.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print('%-12s %s  mean %.3f' % (k, ' '.join('%.3f' % x for x in v), sum(v) / len(v)))
PY
