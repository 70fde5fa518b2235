#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call19
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "small_gemm or folded or resnet18_f32" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED" "$OUT/pytest.log" | head | cut -c1-250
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from simclr_amd import ops
for (M,N,K) in [(512,512,2048),(512,2048,512),(256,256,1024),(256,1024,256),(128,128,512),(64,64,256)]:
    A=torch.randn(M,K,device='cuda'); B=torch.randn(N,K,device='cuda')
    for _ in range(3): ops.small_gemm_nt(A,B)
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.small_gemm_nt(A,B)
    b.record(); torch.cuda.synchronize()
    print('small_gemm %dx%dx%d: %.1f us' % (M,N,K,a.elapsed_time(b)*1e3/20))
PY
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --prof_steps 0"
for v in a b; do
timeout 200 $B > "$OUT/bench_$v.json" 2> "$OUT/bench.err"
done
for f in a b; do
python - "$OUT/bench_$f.json" $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
