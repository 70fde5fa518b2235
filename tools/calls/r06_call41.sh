#!/bin/bash
# round 6, call 41: rolling stem forward without the k-offset table and with 32-bit byte offsets (no spills, no vmcnt(0) at the top of a tile) -- tests, A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call41
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "stem or resnet18 or r18 or cifar or batch32 or reference_source_fixtures" > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3; do
  env SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_a.so timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('stem_conv_fwd', {}).get('ms_per_step'), (d.get('speed_mode') or {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
