#!/bin/bash
# round 5, call 6: what the driver runs at round end, on the final tree: the whole GPU suite (-x), smoke(), and bench.py with the driver's flags
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call6
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1300 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "suite wall seconds: $((SECONDS - T0))" | tee "$OUT/pytest_gpu.time"
tail -3 "$OUT/pytest_gpu.log" | cut -c1-200
T1=$SECONDS
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"; echo "smoke: $((SECONDS - T1)) s"
T2=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench.err"; cut -c1-200 "$OUT/bench_driver_flags.json" | tail -1; echo "bench (driver flags): $((SECONDS - T2)) s"
