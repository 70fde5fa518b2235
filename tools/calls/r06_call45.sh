#!/bin/bash
# round 6, call 45: workgroup target of the nine-tap fp32 weight gradient (SIMCLR_WGRAD3_BLOCKS), per layer
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call45
mkdir -p "$OUT"
cd "$R"
for nb in 256 512 768 1024 1536 2048; do
  SIMCLR_WGRAD3_BLOCKS=$nb timeout 600 python tools/microbench.py --dtype f32 --f32_matmul f16x3_3 --ps --what conv --iters 5 > "$OUT/per_layer_$nb.txt" 2>&1
  echo "== $nb"; grep "k3 s1" "$OUT/per_layer_$nb.txt" | awk '{print $1, $2, $3, $9}'
done
