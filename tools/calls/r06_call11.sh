#!/bin/bash
# round 6, call 11: NT-Xent (separate fwd / bwd plans) gates + microbench; one-rank forced-collectives kernel trace (overlap of RCCL with compute)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call11
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ntxent" > "$OUT/pytest_ntx.txt" 2>&1; tail -4 "$OUT/pytest_ntx.txt"
timeout 300 python tools/microbench.py --what ntxent --iters 30 2>/dev/null | tee "$OUT/ntxent_micro.txt" | tail -8
cd /tmp && export TMPDIR=/tmp
SIMCLR_FORCE_COLLECTIVES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/fc" -o fc -- python "$R/bench.py" --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events > "$OUT/bench_fc.json" 2> "$OUT/fc_err.txt"
cd "$R"
tail -2 "$OUT/fc_err.txt"
python tools/overlap_trace.py "$OUT/fc" --steps 5 --out "$OUT/overlap_forced_collectives.json" | tail -40
python - <<PY
import json
d = json.loads(open('$OUT/bench_fc.json').read().strip().splitlines()[-1])
print('forced collectives: ms', d['ms_per_step'], 'allgather', d.get('allgather'))
PY
