#!/bin/bash
# round 6, call 10: split-fp16 NT-Xent -- gates; workgroup-count sweep of the sweeps' plan
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call10
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ntxent" > "$OUT/pytest_ntx.txt" 2>&1; tail -8 "$OUT/pytest_ntx.txt"
for w in 512 256 384 768 1024; do
  echo "== SIMCLR_NTX_WGS=$w"; SIMCLR_NTX_WGS=$w timeout 300 python tools/microbench.py --what ntxent --iters 30 2>/dev/null | grep -v "n=256\|4096 N=4096"
done > "$OUT/ntxent_sweep.txt"
cat "$OUT/ntxent_sweep.txt"
