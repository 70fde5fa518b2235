#!/bin/bash
# round 4, call 14: fused BatchNorm-backward-reduce dgrad epilogue with mask mode / accumulate compiled in (EPS instantiations):
# parity subset + interleaved A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call14
mkdir -p "$OUT"
cd "$R"
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dgrad_with_fused_bn or folded or test_train_step_bf16 or determinis or bench_path_shapes or fused_conv3 or split_tail" > "$OUT/pytest.log" 2>&1
tail -2 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -10 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1"
for i in 1 2; do
  SIMCLR_BNEPI_SPECIAL=0 timeout 200 $B > "$OUT/generic_$i.json" 2> "$OUT/generic_$i.err"
  timeout 200 $B > "$OUT/special_$i.json" 2> "$OUT/special_$i.err"
done
python - <<'EOP'
import json,os,glob
o=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call14/'
for f in sorted(glob.glob(o+'*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get('kernels',{})
        print('%-18s %8.3f ms %8.1f img/s  fwd %.2f dgrad %.2f wgrad %.2f' % (os.path.basename(f), d['ms_per_step'], d['value'], k.get('conv_igemm_fwd',{}).get('ms_per_step',0), k.get('conv_igemm_dgrad',{}).get('ms_per_step',0), k.get('conv_wgrad',{}).get('ms_per_step',0)))
    except Exception as e: print(os.path.basename(f),'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
