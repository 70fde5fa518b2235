#!/bin/bash
# round 5, call 13: the candidate new defaults (wide tile for the data gradient only, non-temporal BatchNorm kernels, 64-wide tile for K <= 64 only)
# confirmed on cfg2, checked on cfg4 (where the wide forward tile won 2 % in round 4) and on the parity mode (fp32 BatchNorm kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call13
mkdir -p "$OUT"
cd "$R"
C="--no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
run() { name=$1; shift; args=$1; shift; env "$@" timeout 300 python bench.py $C $args > "$OUT/$name.json" 2>> "$OUT/err.txt"; }
for rep in 1 2 3; do
  run cfg2_default_$rep "--steps 12 --warmup 4" A=1
  run cfg2_new3_$rep "--steps 12 --warmup 4" SIMCLR_IGEMM_WIDE=4 SIMCLR_BN_CFG=1 SIMCLR_IGEMM_BN64_K=64
done
K4="--resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3"
for rep in 1 2; do
  run cfg4_default_$rep "$K4" A=1
  run cfg4_wide4_$rep "$K4" SIMCLR_IGEMM_WIDE=4
  run cfg4_bn1k64_$rep "$K4" SIMCLR_BN_CFG=1 SIMCLR_IGEMM_BN64_K=64
  run cfg4_new3_$rep "$K4" SIMCLR_IGEMM_WIDE=4 SIMCLR_BN_CFG=1 SIMCLR_IGEMM_BN64_K=64
done
P="--dtype f32 --f32_matmul bf16x6_3 --steps 6 --warmup 2"
for rep in 1 2; do
  run parity_default_$rep "$P" A=1
  run parity_bn1_$rep "$P" SIMCLR_BN_CFG=1
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
for k, v in sorted(rows.items()):
    print('%-16s %s  mean %.3f' % (k, ' '.join('%.3f' % x for x in v), sum(v) / len(v)))
PY
tail -2 "$OUT/err.txt" | cut -c1-200
