#!/bin/bash
# round 4, call 11: pivoted BatchNorm statistics of the fp32 mode: kernel checks, the fp32 train-step tests (no regression),
# the two step cases VERDICT r03 item 8 names with the pivot on / off, exact-fp32 / parity-mode step time with the pivot
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call11
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "pivoted or presplit or test_conv_fwd_dgrad_wgrad or fixed_thresholds or fast_parity or iid_noise or resnet18_f32 or reference_init or resnet50_sk_f32 or determinis or test_batch_norm or fwd_bn_apply" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  \|pivoted_bn" "$OUT/pytest.log" | head -40 | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_distributed.py -m gpu -q -k "global_batch_oracle or peer_mapped" > "$OUT/pytest_dist.log" 2>&1; tail -2 "$OUT/pytest_dist.log" | cut -c1-200
timeout 900 python tools/pivot_report.py --r152 --randbn --out "$OUT/pivot_report.json" > "$OUT/pivot_report.log" 2>&1; grep -v "^ok   kernel" "$OUT/pivot_report.log" | grep "OVER\|step_\|fixed_" | cut -c1-230 | head -60
grep "kernel" "$OUT/pivot_report.log" | cut -c1-250 | head -20
P="python bench.py --dtype f32 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --no_kernel_events"
SIMCLR_BN_PIVOT=0 timeout 200 $P --f32_matmul bf16x6_3 > "$OUT/par_raw.json" 2> "$OUT/par_raw.err"
timeout 200 $P --f32_matmul bf16x6_3 > "$OUT/par_pivot.json" 2> "$OUT/par_pivot.err"
python - <<'EOP'
import json,os,glob
o=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call11/'
for f in sorted(glob.glob(o+'par_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']))
    except Exception as e: print(os.path.basename(f),'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
