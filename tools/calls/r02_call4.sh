#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call4
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "augmentation or fixed_thresholds or iid_noise or 2x_sk" > "$OUT/pytest.log" 2>&1
tail -8 "$OUT/pytest.log"; grep -n "fixed_\|grad-error share\|FAILED\|Error\|augment" "$OUT/pytest.log" | cut -c1-220 | head -70
timeout 600 python tools/bf16_parity_report.py > "$OUT/bf16_report.log" 2>&1; grep -v "grad-error" "$OUT/bf16_report.log" | tail -40
cp gpurun_out/bf16_parity.json "$OUT/" 2>/dev/null
