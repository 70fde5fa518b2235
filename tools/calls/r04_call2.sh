#!/bin/bash
# round 4, call 2: split tail of the persistent forward / dgrad grid (parity + A/B in the step), kernel breakdown of the fast parity mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call2
mkdir -p "$OUT"
cd "$R"
timeout 700 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split_tail or bench_path or 256_tile or dgrad_with_fused or conv_fwd_dgrad_wgrad or bitwise or determinis or test_train_step_bf16" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run split_a X=1
run nosplit_a SIMCLR_IGEMM_SPLIT=0
run split_b X=1
run nosplit_b SIMCLR_IGEMM_SPLIT=0
timeout 200 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --dtype f32 --f32_matmul bf16x6_3 > "$OUT/bench_parity.json" 2> "$OUT/bench_parity.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call2/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-28s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), {k: v.get('ms_per_step') if isinstance(v, dict) else v for k, v in (d.get('kernels') or {}).items()} if 'parity' in f else d['roofline'].get('frac'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
