#!/bin/bash
# round 4, call 4 (= call 3 after the partial-store fix): eight-phase wide tile (parity + per-layer A/B), split tail with write-through partials (parity + A/B in the step)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call4
mkdir -p "$OUT"
cd "$R"
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wide_eight_phase" > "$OUT/pytest_wide.log" 2>&1
tail -3 "$OUT/pytest_wide.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_wide.log" | head -20 | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split_tail or with_wide_tiles" > "$OUT/pytest_split.log" 2>&1
tail -3 "$OUT/pytest_split.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_split.log" | head -20 | cut -c1-250
timeout 400 python tools/microbench.py --what wide --iters 5 --out "$OUT/wide.json" > "$OUT/wide.log" 2>&1
tail -25 "$OUT/wide.log" | cut -c1-200
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run nosplit_a SIMCLR_IGEMM_SPLIT=0
run split_a X=1
run wide_a SIMCLR_IGEMM_WIDE=1
run nosplit_b SIMCLR_IGEMM_SPLIT=0
run split_b X=1
run wide_b SIMCLR_IGEMM_WIDE=1
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call4/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-28s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['roofline'].get('frac'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
