#!/bin/bash
# round 3, call 4: parity of the pieces changed since call 3 (unrolled stem, batched metric bookkeeping, nine-tap split defaults),
# interleaved bench A/B: stem unroll, epilogue start skew, the three 256-wide tile classes; depth-152 test on image-like inputs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call4
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multitap or stem_conv or test_train_step_bf16 or determinis or hand_derived or supervised or model_api or checkpoint_resume or resnet18_f32" > "$OUT/pytest_a.log" 2>&1
tail -3 "$OUT/pytest_a.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_a.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run base_a X=1
run nounroll SIMCLR_STEM_UNROLL=0
run skew05 SIMCLR_EPI_SKEW=0.5
run skew1 SIMCLR_EPI_SKEW=1
run skew2 SIMCLR_EPI_SKEW=2
run base_b X=1
run c1 SIMCLR_IGEMM_256_CLASSES=1
run c2 SIMCLR_IGEMM_256_CLASSES=2
run c4 SIMCLR_IGEMM_256_CLASSES=4
run base_c X=1
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call4/bench_*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-22s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), d['step_ms'], d['train_metrics'] if 'base_a' in f else '')
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
EOP
timeout 100 python tools/find_fills.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | tee "$OUT/find_fills.txt" | head -30
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "resnet152" > "$OUT/pytest_152.log" 2>&1
tail -3 "$OUT/pytest_152.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_152.log" | head -20 | cut -c1-250
