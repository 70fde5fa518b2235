#!/bin/bash
# round 4, call 10: parity-mode speed-ups (pre-split weights for the three-term dgrad, in-LDS split for the wgrad) and the
# SIMCLR_CONV3_EPI=preapply switch: parity subset, interleaved A/B timings, bf16 drift of the fused-tail variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call10
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "presplit or split_bf16 or fwd_bn_apply or fast_parity" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -20 | cut -c1-250
P="python bench.py --dtype f32 --f32_matmul bf16x6_3 --steps 10 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 2"
for i in 1 2; do
  SIMCLR_F32_PRESPLIT=0 timeout 200 $P > "$OUT/par_off_$i.json" 2> "$OUT/par_off_$i.err"
  timeout 200 $P > "$OUT/par_on_$i.json" 2> "$OUT/par_on_$i.err"
done
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1"
for i in 1 2; do
  SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_nopre.so timeout 200 $B > "$OUT/bf16_nopre_$i.json" 2> "$OUT/bf16_nopre_$i.err"
  timeout 200 $B > "$OUT/bf16_new_$i.json" 2> "$OUT/bf16_new_$i.err"
done
SIMCLR_CONV3_EPI=preapply timeout 200 $B > "$OUT/bf16_preapply.json" 2> "$OUT/bf16_preapply.err"
python - <<'EOP'
import json,os,glob
o=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call10/'
for f in sorted(glob.glob(o+'*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get('kernels',{})
        print('%-22s %8.3f ms %8.1f img/s  fwd %.2f dgrad %.2f wgrad %.2f' % (os.path.basename(f), d['ms_per_step'], d['value'], k.get('conv_igemm_fwd',{}).get('ms_per_step',0), k.get('conv_igemm_dgrad',{}).get('ms_per_step',0), k.get('conv_wgrad',{}).get('ms_per_step',0)))
    except Exception as e: print(os.path.basename(f),'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
timeout 600 python tools/step_modes.py --modes bf16,bf16@preapply,bf16@unfused --out "$OUT/step_modes_bf16.json" > "$OUT/step_modes_bf16.log" 2>&1; tail -12 "$OUT/step_modes_bf16.log" | cut -c1-200
