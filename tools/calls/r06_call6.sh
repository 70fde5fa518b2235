#!/bin/bash
# round 6, call 6: fused fp32 tail with batched residual loads + 4-byte mask stores, split-fp16 stem; policy sweep for the fp32 kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call6
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_optimizers.py -q -m gpu -k "fused_bn_apply_tail_f32 or stem_conv or optimizers or sgd_adam or fast_parity_mode" > "$OUT/pytest_sel.txt" 2>&1
tail -6 "$OUT/pytest_sel.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
run() { tag=$1; shift; env "$@" timeout 300 $B > "$OUT/bench_$tag.json" 2>> "$OUT/err.txt"; }
for rep in 1 2; do
  run default_$rep X=1
  run unfused_$rep SIMCLR_CONV3_FUSED_F32=0
  run stem6_$rep SIMCLR_STEM_F16=0
done
run wg1024 SIMCLR_WGRAD_BLOCKS=1024
run wg2048 SIMCLR_WGRAD_BLOCKS=2048
run wg3072 SIMCLR_WGRAD_BLOCKS=3072
run bncfg0 SIMCLR_BN_CFG=0
run wgstream SIMCLR_WGRAD_STREAM=1
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if v.get('ms_per_step', 0) > 1.0})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -5 "$OUT/err.txt"
