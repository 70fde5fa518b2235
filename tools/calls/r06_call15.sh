#!/bin/bash
# round 6, call 15: two-replica step in the default fast parity mode; A/B of three resident workgroups per CU for the fp32 64-wide three-term tiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call15
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "fast_parity_mode or two_replica_step_equals" > "$OUT/pytest_dist.txt" 2>&1; tail -4 "$OUT/pytest_dist.txt"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split_bf16_matmul or presplit_gradient or split_bf16_bench_path" > "$OUT/pytest_k.txt" 2>&1; tail -2 "$OUT/pytest_k.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2"
for rep in 1 2 3; do
  SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_a.so timeout 300 $B > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if v.get('ms_per_step', 0) > 1.0})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
