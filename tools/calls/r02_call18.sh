#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call18
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "resnet18_f32 or deterministic or prep" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --prof_steps 0"
for v in a b; do
timeout 200 $B > "$OUT/bench_def_$v.json" 2> "$OUT/bench.err"
SIMCLR_PREP_BATCH=0 timeout 200 $B > "$OUT/bench_prep0_$v.json" 2> "$OUT/bench_off.err"
SIMCLR_WGRAD_BLOCKS=768 timeout 200 $B > "$OUT/bench_wb768_$v.json" 2> "$OUT/bench_off.err"
SIMCLR_WGRAD_BLOCKS=1024 timeout 200 $B > "$OUT/bench_wb1024_$v.json" 2> "$OUT/bench_off.err"
SIMCLR_WGRAD_BLOCKS=2304 timeout 200 $B > "$OUT/bench_wb2304_$v.json" 2> "$OUT/bench_off.err"
done
for f in def_a prep0_a wb768_a wb1024_a wb2304_a def_b prep0_b wb768_b wb1024_b wb2304_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench.err"
