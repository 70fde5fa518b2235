#!/bin/bash
# round 3, call 9: three-stage LDS ring of the nine-tap 3x3 weight gradient (W <= 28 layers): parity, stand-alone and in-step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call9
mkdir -p "$OUT"
cd "$R"
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multitap or (bench_path and k3) or bench_path_shapes" > "$OUT/pytest.log" 2>&1
tail -2 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -10 | cut -c1-250
echo "--- 3 stages"; timeout 100 python tools/bench_wgrad3x3.py 2>&1 | grep -v amdgpu.ids
echo "--- 2 stages"; SIMCLR_WGRAD3_STAGES=2 timeout 100 python tools/bench_wgrad3x3.py 2>&1 | grep -v amdgpu.ids
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run s3_a X=1
run s2_a SIMCLR_WGRAD3_STAGES=2
run s3_b X=1
run s2_b SIMCLR_WGRAD3_STAGES=2
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call9/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['kernels']['conv_wgrad'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
