#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call15
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bench_path and dtype0" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_alt.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bench_path and dtype0 and 64-64-3" > "$OUT/pytest_alt.log" 2>&1
tail -3 "$OUT/pytest_alt.log" | cut -c1-250
for w in pin nopin wres0; do
L=$R/simclr_amd/libsimclr_hip.so; [ $w = nopin ] && L=$R/simclr_amd/libsimclr_hip_alt.so
E=1; [ $w = wres0 ] && E=0
SIMCLR_CONV3_WRES=$E SIMCLR_HIP_LIB=$L timeout 300 python tools/microbench.py --what conv --out "$OUT/micro_$w.json" > "$OUT/micro_$w.log" 2>&1
echo "== $w"; grep "56x56 64->64 k3\|totals" "$OUT/micro_$w.log"
done
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_pin_$v.json" 2> "$OUT/bench_pin.err"
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_alt.so timeout 200 $B > "$OUT/bench_nopin_$v.json" 2> "$OUT/bench_nopin.err"
SIMCLR_CONV3_WRES=0 timeout 200 $B > "$OUT/bench_wres0_$v.json" 2> "$OUT/bench_wres0.err"
done
for f in pin_a nopin_a wres0_a pin_b nopin_b wres0_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench_pin.err"
