#!/bin/bash
# round 5, call 2: the whole GPU suite with per-test durations (900 s of a 1200 s limit -> find what to shrink)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call2
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1500 python -m pytest tests -m gpu -q --durations=120 > "$OUT/pytest_gpu.log" 2>&1
echo "suite wall seconds: $((SECONDS - T0))" | tee "$OUT/pytest_gpu.time"
grep -n "passed\|failed" "$OUT/pytest_gpu.log" | tail -3
grep -n "^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | head -20
grep -A125 "slowest 120 durations" "$OUT/pytest_gpu.log" | head -70 | cut -c1-200
