#!/bin/bash
# round 4, call 5: the wide-tile rule as default (parity subset) + A/B in the step (cfg2 and cfg4)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call5
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wide or split_tail or bitwise or bench_path_shapes or 256_tile or determinis or test_train_step_bf16 or fixed_thresholds" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run default_a X=1
run nowide_a SIMCLR_IGEMM_WIDE=0
run default_b X=1
run nowide_b SIMCLR_IGEMM_WIDE=0
run widenosplit_a SIMCLR_IGEMM_SPLIT=0
C4="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_f32 --no_kernel_events --width_multiplier 2 --sk_ratio 0.0625"
env X=1 timeout 300 $C4 > "$OUT/bench_cfg4_default.json" 2> "$OUT/bench_cfg4_default.err"
env SIMCLR_IGEMM_WIDE=0 timeout 300 $C4 > "$OUT/bench_cfg4_nowide.json" 2> "$OUT/bench_cfg4_nowide.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call5/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-28s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), (d.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
