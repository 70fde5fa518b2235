#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call27
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "stem or gram or deterministic or resnet18_f32 or fused_conv3 or free_proj" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_new_$v.json" 2> "$OUT/bench.err"
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_prev.so timeout 200 $B > "$OUT/bench_prev_$v.json" 2> "$OUT/bench_prev.err"
done
for f in new_a prev_a new_b prev_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'], d['kernels']['stem_conv_fwd'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench.err"
