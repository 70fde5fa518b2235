#!/bin/bash
# round 3, call 12: the whole GPU suite + smoke on the final commit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call12
mkdir -p "$OUT"
cd "$R"
timeout 790 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_gpu.log" | head
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
