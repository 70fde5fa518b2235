#!/bin/bash
# round 4, call 18: the two device tests against the reference-source fixtures (tests/golden/reference_pin.npz)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call18
mkdir -p "$OUT"
cd "$R"
timeout 60 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "reference_source_fixtures" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT"/pytest.log | head -12 | cut -c1-300
