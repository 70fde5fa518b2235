#!/bin/bash
# round 4, call 1: split-bf16 fp32 matrix arithmetic (parity + speed), vendor comparator table, this box's baseline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call1
mkdir -p "$OUT"
cd "$R"
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split_bf16" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -20 | cut -c1-250
timeout 500 python tools/step_modes.py --out "$OUT/step_modes.json" > "$OUT/step_modes.log" 2>&1
tail -14 "$OUT/step_modes.log" | cut -c1-200
B="python bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_kernel_events"
for m in exact bf16x3 bf16x6_3 bf16x6; do
  timeout 200 $B --dtype f32 --f32_matmul $m > "$OUT/bench_f32_$m.json" 2> "$OUT/bench_f32_$m.err"
done
timeout 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call1/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-28s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
timeout 420 python tools/microbench.py --what comparator --iters 5 --out "$OUT/comparator.json" > "$OUT/comparator.log" 2>&1
tail -30 "$OUT/comparator.log" | cut -c1-200
