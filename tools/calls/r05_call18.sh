#!/bin/bash
# round 5, call 18: the whole GPU suite and smoke() on the final tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call18
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "suite wall seconds: $((SECONDS - T0))"; tail -2 "$OUT/pytest_gpu.log" | cut -c1-200; grep -n "^FAILED\|^E  " "$OUT/pytest_gpu.log" | head -10 | cut -c1-250
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
