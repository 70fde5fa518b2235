#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call13
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bench_path" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head -20 | cut -c1-250
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_wpe3.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bench_path and dtype0" > "$OUT/pytest_wpe3.log" 2>&1
tail -3 "$OUT/pytest_wpe3.log" | cut -c1-250
for w in def wpe3; do
L=$R/simclr_amd/libsimclr_hip.so; [ $w = wpe3 ] && L=$R/simclr_amd/libsimclr_hip_wpe3.so
SIMCLR_HIP_LIB=$L timeout 300 python tools/microbench.py --what conv --out "$OUT/micro_$w.json" > "$OUT/micro_$w.log" 2>&1
echo "== $w"; grep "k3 s1\|56x56 64\|56x56 256->64\|totals" "$OUT/micro_$w.log"
done
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for v in a b; do
timeout 200 $B > "$OUT/bench_def_$v.json" 2> "$OUT/bench_def.err"
SIMCLR_HIP_LIB=$R/simclr_amd/libsimclr_hip_wpe3.so timeout 200 $B > "$OUT/bench_wpe3_$v.json" 2> "$OUT/bench_wpe3.err"
done
for f in def_a wpe3_a def_b wpe3_b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 "$OUT/bench_wpe3.err"
