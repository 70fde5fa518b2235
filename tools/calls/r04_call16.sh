#!/bin/bash
# round 4, call 16: safety subset on the final build (bf16 step tests that use the specialised epilogues, two-replica tests)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call16
mkdir -p "$OUT"
cd "$R"
timeout 150 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "fused_tail or without_global_bn" > "$OUT/pytest_dist.log" 2>&1; tail -2 "$OUT/pytest_dist.log" | cut -c1-200
timeout 110 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "2x_sk_randomized or wide_tiles or 256_tiles or free_proj" > "$OUT/pytest.log" 2>&1; tail -2 "$OUT/pytest.log" | cut -c1-200; grep -n "^FAILED\|^E  " "$OUT"/pytest*.log | head -10 | cut -c1-250
