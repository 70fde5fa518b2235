#!/bin/bash
# round 5, call 7: the product's backward against central differences of the reference's own training step (new fixture step_r18_img_R1_grad_fd)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call7
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "single_step_matches or bf16_speed_mode" > "$OUT/pytest.log" 2>&1
grep -n "backward_vs\|passed\|failed" "$OUT/pytest.log" | cut -c1-220 | tail -14
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"; grep -n "backward_vs" "$OUT/smoke.log" | cut -c1-200
