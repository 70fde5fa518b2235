#!/bin/bash
# round 2, GPU call 1: new bench-path parity tests + bench A/B (wgrad side stream, multi-tap wgrad)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call1
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "bench_path or fixed_thresholds" > "$OUT/pytest_new.log" 2>&1
tail -25 "$OUT/pytest_new.log"
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
SIMCLR_WGRAD_STREAM=1 timeout 200 $B > "$OUT/bench_stream.json" 2> "$OUT/bench_stream.err"
SIMCLR_WGRAD_3X3=2 timeout 200 $B > "$OUT/bench_3x3.json" 2> "$OUT/bench_3x3.err"
SIMCLR_WGRAD_STREAM=1 SIMCLR_WGRAD_3X3=2 timeout 200 $B > "$OUT/bench_stream_3x3.json" 2> "$OUT/bench_stream_3x3.err"
for f in base stream 3x3 stream_3x3; do
  python - "$OUT/bench_$f.json" "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
timeout 400 python bench.py --steps 10 --warmup 3 > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -c 3000 "$OUT/bench_full.json"; tail -5 "$OUT/bench_full.err"
