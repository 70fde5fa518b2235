#!/bin/bash
# round 6, call 14: distributed tests after the comm changes; cfg4 / cfg5 on one GPU in the headline mode and in the bf16 speed mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call14
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu > "$OUT/pytest_dist.txt" 2>&1; tail -3 "$OUT/pytest_dist.txt"
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
timeout 600 $B --width_multiplier 2 --sk_ratio 0.0625 --per_gpu_batch 256 > "$OUT/cfg4_b256_parity.json" 2>> "$OUT/err.txt"
timeout 600 $B --width_multiplier 2 --sk_ratio 0.0625 --per_gpu_batch 512 > "$OUT/cfg4_b512_parity.json" 2>> "$OUT/err.txt"
timeout 600 $B --width_multiplier 2 --sk_ratio 0.0625 --per_gpu_batch 512 --dtype bf16 > "$OUT/cfg4_b512_bf16.json" 2>> "$OUT/err.txt"
timeout 900 $B --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 128 > "$OUT/cfg5_b128_parity.json" 2>> "$OUT/err.txt"
timeout 900 $B --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --dtype bf16 > "$OUT/cfg5_b256_bf16.json" 2>> "$OUT/err.txt"
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/cfg*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], 'mfma_frac', d['step_mfma_frac'], 'peak_hbm_gb', d['peak_hbm_gb'], {n: v['ms_per_step'] for n, v in d['kernels'].items() if v['ms_per_step'] > 5})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -5 "$OUT/err.txt"
