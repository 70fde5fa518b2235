#!/bin/bash
# round 2, GPU call 2: whole GPU suite on the new build (deterministic statistics, batched SyncBN, async collective A,
# hand-derived cases, fixed-threshold steps incl. fp32-head variant), serialized-kernel run of the conv tests, bench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call2
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -q -s -x > "$OUT/pytest_full.log" 2>&1
tail -5 "$OUT/pytest_full.log"; grep -n "fixed_\|grad-error share" "$OUT/pytest_full.log" | head -80
# conv / BN kernels with serialized launches and an un-cached allocator (every tensor its own
# hipMalloc: an out-of-bounds access lands outside the allocation instead of in a neighbouring cached block)
AMD_SERIALIZE_KERNEL=3 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x \
  -k "conv or dgrad or batch_norm or stem or pooling" > "$OUT/pytest_serialized.log" 2>&1
tail -3 "$OUT/pytest_serialized.log"
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
timeout 200 $B > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
python - "$OUT/bench_base.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], d['step_ms'], d['ntxent'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
