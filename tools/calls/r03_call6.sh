#!/bin/bash
# round 3, call 6: mode-4 dgrad epilogue with early operand loads, 64-wide tiles (3 workgroups per CU) for the short-K streaming
# layers: parity under both switches, interleaved bench A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call6
mkdir -p "$OUT"
cd "$R"
K="fused_bn_backward or fwd_bn_apply or bench_path or test_train_step_bf16 or fused_conv3"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "$K" > "$OUT/pytest_a.log" 2>&1
tail -2 "$OUT/pytest_a.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_a.log" | head -20 | cut -c1-250
SIMCLR_IGEMM_BN64_K=128 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "$K" > "$OUT/pytest_b.log" 2>&1
tail -2 "$OUT/pytest_b.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_b.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run base_a X=1
run nom4 SIMCLR_DGRAD_M4=0
run n64 SIMCLR_IGEMM_BN64_K=64
run n128 SIMCLR_IGEMM_BN64_K=128
run n256 SIMCLR_IGEMM_BN64_K=256
run base_b X=1
run nom4_b SIMCLR_DGRAD_M4=0
run n128_b SIMCLR_IGEMM_BN64_K=128
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call6/bench_*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['step_ms'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if k.startswith('conv_igemm')})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
