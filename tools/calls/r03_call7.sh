#!/bin/bash
# round 3, call 7: time attribution of the current conv kernels (diagnostic library), cfg4 with the wide tile classes, bench of
# the cleaned-up defaults
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call7
mkdir -p "$OUT"
cd "$R"
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_diag.so timeout 300 python tools/diag_conv.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/diag_conv.txt"
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
C4="python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 6 --warmup 2 --no_cpu_baseline --no_f32 --prof_steps 1"
timeout 300 $C4 > "$OUT/cfg4_base.json" 2> "$OUT/cfg4_base.err"
SIMCLR_IGEMM_256_CLASSES=6 timeout 300 $C4 > "$OUT/cfg4_c6.json" 2> "$OUT/cfg4_c6.err"
SIMCLR_IGEMM_256_CLASSES=7 timeout 300 $C4 > "$OUT/cfg4_c7.json" 2> "$OUT/cfg4_c7.err"
SIMCLR_IGEMM_BN64_K=0 timeout 300 $C4 > "$OUT/cfg4_n0.json" 2> "$OUT/cfg4_n0.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call7/*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), {k: v['ms_per_step'] for k, v in d['kernels'].items() if k.startswith('conv')})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
