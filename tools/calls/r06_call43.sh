#!/bin/bash
# round 6, call 43: s_setprio(1) around the MFMA cluster of the weight gradient on a pre-split gradient only -- five interleaved pairs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call43
mkdir -p "$OUT"
cd "$R"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32"
for rep in 1 2 3 4 5; do
  env SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_a.so timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d['kernels']
        print(os.path.basename(f), d['ms_per_step'], k.get('conv_igemm_fwd', {}).get('ms_per_step'), k.get('conv_igemm_dgrad', {}).get('ms_per_step'), k.get('conv_wgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
