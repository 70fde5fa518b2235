#!/bin/bash
# round 3, call 8: NT-Xent with register-prefetched tiles: parity + stand-alone timing against the previous library; defaults check
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call8
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "ntxent or free_proj_out" > "$OUT/pytest_nt.log" 2>&1
tail -2 "$OUT/pytest_nt.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_nt.log" | head -10 | cut -c1-250
echo "--- new"; timeout 100 python tools/microbench.py --what ntxent --iters 9 --out "$OUT/nt_new.json" 2>&1 | grep -v amdgpu.ids
echo "--- old"; SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_ntold.so timeout 100 python tools/microbench.py --what ntxent --iters 9 --out "$OUT/nt_old.json" 2>&1 | grep -v amdgpu.ids
echo "--- new"; timeout 100 python tools/microbench.py --what ntxent --iters 9 --out "$OUT/nt_new2.json" 2>&1 | grep -v amdgpu.ids
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
SIMCLR_IGEMM_256_CLASSES=0 timeout 200 $B > "$OUT/bench_c0.json" 2> "$OUT/bench_c0.err"
timeout 200 $B > "$OUT/bench_base_b.json" 2> "$OUT/bench_base_b.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call8/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['ntxent'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
