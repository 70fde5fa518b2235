#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call29
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_kernels.py -m gpu -q -k "bf16_resnet50_fused or fused_conv3 or (deterministic and 50)" > "$OUT/pytest.log" 2>&1
tail -2 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head | cut -c1-300
timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-230 "$OUT/bench.json" | tail -1
