#!/bin/bash
# round 3, call 10: stem weight gradient over LDS-DMA (all seven kernel rows in one 256-row k-tile): parity + A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call10
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "stem or resnet18_f32 or test_train_step_bf16 or determinis" > "$OUT/pytest.log" 2>&1
tail -2 "$OUT/pytest.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest.log" | head -10 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run dma_a X=1
run old_a SIMCLR_STEM_WGRAD_DMA=0
run dma_b X=1
run old_b SIMCLR_STEM_WGRAD_DMA=0
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call10/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['kernels']['conv_wgrad'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
