#!/bin/bash
# round 5, call 4: the whole GPU suite after the shrink (target <= 720 s; round 4: 900 s, call 2 of this round: 1007 s) + the new split-bf16
# stem kernels, then the parity mode with and without them (interleaved, one box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call4
mkdir -p "$OUT"
cd "$R"
T0=$SECONDS
timeout 1300 python -m pytest tests -m gpu -q -x --durations=25 > "$OUT/pytest_gpu.log" 2>&1
echo "suite wall seconds: $((SECONDS - T0))" | tee "$OUT/pytest_gpu.time"
grep -n "passed\|failed" "$OUT/pytest_gpu.log" | tail -3
grep -n "^FAILED\|^ERROR\|^E  " "$OUT/pytest_gpu.log" | head -30 | cut -c1-300
grep -A30 "slowest 25 durations" "$OUT/pytest_gpu.log" | head -30 | cut -c1-160
P="python bench.py --dtype f32 --f32_matmul bf16x6_3 --steps 10 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --prof_steps 2"
for i in 1 2; do
  SIMCLR_STEM_SPLIT=0 timeout 200 $P > "$OUT/parity_exactstem_$i.json" 2> "$OUT/err.txt"
  timeout 200 $P > "$OUT/parity_splitstem_$i.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json
for n in ('parity_exactstem_1', 'parity_splitstem_1', 'parity_exactstem_2', 'parity_splitstem_2'):
    try:
        d = json.loads(open('$OUT/' + n + '.json').read().strip().splitlines()[-1])
        k = d['kernels']
        print(n, d['value'], d['ms_per_step'], 'stem fwd', k.get('stem_conv_fwd', {}).get('ms_per_step'), 'wgrad', k.get('conv_wgrad', {}).get('ms_per_step'))
    except Exception as e:
        print(n, 'failed', e)
PY
tail -3 "$OUT/err.txt" | cut -c1-300
