#!/bin/bash
# round 5, call 12: focused, three-fold interleaved A/B of the switches call 11 flagged: the eight-phase wide tile (off / forward only / dgrad only),
# the non-temporal policy of the BatchNorm kernels, the 64-wide tile threshold -- alone and combined
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call12
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/$name.json" 2>> "$OUT/err.txt"; }
for rep in 1 2 3; do
  run default_$rep A=1
  run wide0_$rep SIMCLR_IGEMM_WIDE=0
  run widefwd_$rep SIMCLR_IGEMM_WIDE=3
  run widedgrad_$rep SIMCLR_IGEMM_WIDE=4
  run bncfg1_$rep SIMCLR_BN_CFG=1
  run bn64k64_$rep SIMCLR_IGEMM_BN64_K=64
  run wide0bn1_$rep SIMCLR_IGEMM_WIDE=0 SIMCLR_BN_CFG=1
  run all3_$rep SIMCLR_IGEMM_WIDE=0 SIMCLR_BN_CFG=1 SIMCLR_IGEMM_BN64_K=64
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
