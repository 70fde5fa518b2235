#!/bin/bash
# round 3, call 11: 64 x 64 tile (4 workgroups per CU) for the big-output short-K streaming layers: parity + A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call11
mkdir -p "$OUT"
cd "$R"
K="fused_bn_backward or fwd_bn_apply or bench_path or test_train_step_bf16 or fused_conv3"
SIMCLR_IGEMM_T64_K=128 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "$K" > "$OUT/pytest_t64.log" 2>&1
tail -2 "$OUT/pytest_t64.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_t64.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run base_a X=1
run t64_a SIMCLR_IGEMM_T64_K=64
run t128_a SIMCLR_IGEMM_T64_K=128
run base_b X=1
run t64_b SIMCLR_IGEMM_T64_K=64
run t128_b SIMCLR_IGEMM_T64_K=128
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call11/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), {k: v['ms_per_step'] for k, v in d['kernels'].items() if k.startswith('conv_igemm')})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
