#!/bin/bash
# round 3, call 5: parity + A/B of the FAPPLY residual prefetch / packed mask stores and the branch-free max-pool backward;
# BASELINE configs[3] and [4] (cfg4, cfg5) bench lines on one GPU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call5
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fwd_bn_apply or pooling or fused_conv3 or test_train_step_bf16 or determinis or 256_tile or stem_backward or resnet50_sk" > "$OUT/pytest_a.log" 2>&1
tail -3 "$OUT/pytest_a.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_a.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run base_a X=1
run nopf SIMCLR_FAPPLY_PF=0
run base_b X=1
run nopf_b SIMCLR_FAPPLY_PF=0
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call5/bench_*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['step_ms'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
timeout 400 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-330 "$OUT/bench_cfg4.json" | tail -1; tail -3 "$OUT/bench_cfg4.err" | cut -c1-300
timeout 600 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 128 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --prof_steps 1 > "$OUT/bench_cfg5_b128.json" 2> "$OUT/bench_cfg5_b128.err"; cut -c1-330 "$OUT/bench_cfg5_b128.json" | tail -1; tail -3 "$OUT/bench_cfg5_b128.err" | cut -c1-300
PK=$(python -c "import json;print(json.loads(open('$OUT/bench_cfg5_b128.json').read().strip().splitlines()[-1])['peak_hbm_gb'])" 2>/dev/null || echo 999)
echo "cfg5 peak HBM at 128 images/GPU: $PK GB"
if python -c "import sys; sys.exit(0 if float('$PK') * 2 < 265 else 1)"; then
  timeout 700 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --prof_steps 1 > "$OUT/bench_cfg5_b256.json" 2> "$OUT/bench_cfg5_b256.err"; cut -c1-330 "$OUT/bench_cfg5_b256.json" | tail -1; tail -3 "$OUT/bench_cfg5_b256.err" | cut -c1-300
fi
