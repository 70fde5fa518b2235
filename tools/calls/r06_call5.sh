#!/bin/bash
# round 6, call 5: fp32 fused tail (Gram statistics + conv3 epilogue) -- kernel test, fixed gates, A/B in the step; re-run of call 4's failures
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call5
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_optimizers.py -q -m gpu -k "fused_bn_apply_tail_f32 or presplit_gradient or r50_sk or r152_3x or optimizers or sgd_adam or fast_parity_mode or pin_step or single_step_matches or resnet152" > "$OUT/pytest_sel.txt" 2>&1
tail -12 "$OUT/pytest_sel.txt"
timeout 600 python tools/step_modes.py --modes f16x3_3 --out "$OUT/step_modes.json" > "$OUT/step_modes.txt" 2>&1
tail -10 "$OUT/step_modes.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
for rep in 1 2; do
  SIMCLR_CONV3_FUSED_F32=0 timeout 300 $B > "$OUT/bench_unfused_$rep.json" 2>> "$OUT/err.txt"
  SIMCLR_CONV3_FUSED_F32=1 timeout 300 $B > "$OUT/bench_fused_$rep.json" 2>> "$OUT/err.txt"
done
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
