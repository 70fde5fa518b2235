#!/bin/bash
# round 5, call 15: new defaults against the round-4 policy (reproduced by switches), interleaved on one box: cfg2 x 3, cfg4 x 2; trajectory test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call15
mkdir -p "$OUT"
cd "$R"
C="--no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
run() { name=$1; shift; args=$1; shift; env "$@" timeout 300 python bench.py $C $args > "$OUT/$name.json" 2>> "$OUT/err.txt"; }
OLD="SIMCLR_BN_CFG=0 SIMCLR_IGEMM_BN64_K=128 SIMCLR_IGEMM_WIDE=5 SIMCLR_WGRAD_BLOCKS=1536"
for rep in 1 2 3; do
  run cfg2_old_$rep "--steps 12 --warmup 4" $OLD
  run cfg2_new_$rep "--steps 12 --warmup 4" A=1
done
K4="--resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3"
for rep in 1 2; do
  run cfg4_old_$rep "$K4" $OLD
  run cfg4_new_$rep "$K4" A=1
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
    print('%-10s %s  mean %.3f' % (k, ' '.join('%.3f' % x for x in v), sum(v) / len(v)))
PY
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "trajectory" 2>&1 | grep -i "traj_contrast\|passed\|failed" | cut -c1-220
