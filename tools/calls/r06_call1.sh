#!/bin/bash
# round 6, call 1: split-fp16 forward (f16x3_3) -- fp16 MFMA probe (subnormal inputs), kernel gates, the fixed-gate R50/224 step, and the
# parity-mode step time against bf16x6_3 (interleaved on this box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call1
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "probes or split_bf16_matmul or split_bf16_bench_path" > "$OUT/pytest_kernels.txt" 2>&1
tail -5 "$OUT/pytest_kernels.txt"
timeout 900 python tools/step_modes.py --modes bf16x6_3,f16x3_3 --out "$OUT/step_modes.json" > "$OUT/step_modes.txt" 2>&1
tail -40 "$OUT/step_modes.txt"
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_pmc --no_parity --no_f32 --prof_steps 2 --dtype f32"
for rep in 1 2; do
for m in bf16x6_3 f16x3_3; do
  timeout 300 $B --f32_matmul $m > "$OUT/bench_${m}_$rep.json" 2>> "$OUT/err.txt"
done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        print(os.path.basename(f), d['ms_per_step'], {n: v['ms_per_step'] for n, v in k.items() if v.get('ms_per_step', 0) > 1.0})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -5 "$OUT/err.txt"
