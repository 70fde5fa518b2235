#!/bin/bash
# round 3, call 3: parity of the changed kernels + new soundness / distributed tests, bench A/B on one box (new defaults vs
# 128-wide tiles vs per-tap 3x3 wgrad), nine-tap split sweep, torch-kernel census, LDS counters of the nine-tap kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_call3
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multitap or 256_tile or with_256" > "$OUT/pytest_kernels.log" 2>&1
tail -3 "$OUT/pytest_kernels.log" | cut -c1-300; grep -n "^FAILED\|^E  " "$OUT/pytest_kernels.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32"
for rep in a b; do
  timeout 200 $B > "$OUT/bench_new_$rep.json" 2> "$OUT/bench_new_$rep.err"
  SIMCLR_IGEMM_TILE=128 timeout 200 $B > "$OUT/bench_t128_$rep.json" 2> "$OUT/bench_t128_$rep.err"
  SIMCLR_WGRAD_3X3=0 timeout 200 $B > "$OUT/bench_w0_$rep.json" 2> "$OUT/bench_w0_$rep.err"
done
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['step_ms'])
    except Exception as e: print(f, 'ERR', e)
EOP
timeout 200 python tools/bench_wgrad3x3.py --blocks 512,768,1024,1536,2048 2>&1 | grep -v amdgpu.ids | tee "$OUT/wgrad3_blocks.txt"
timeout 200 python tools/find_fills.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/find_fills.txt" | head -45
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_distributed.py -m gpu -q -s -k "trained_point or trajectory or resnet152 or one_rank or two_ranks_over_gloo or without_global_bn" > "$OUT/pytest_new.log" 2>&1
tail -3 "$OUT/pytest_new.log" | cut -c1-300; grep -n "err=\|^FAILED\|pretrained\|^E  " "$OUT/pytest_new.log" | head -70 | cut -c1-230
cp gpurun_out/bf16_trajectory.json "$OUT/" 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_w3" -o w3 -- python $R/tools/bench_wgrad3x3.py > "$OUT/pmc_w3.log" 2>&1
python - <<'EOP'
import csv,glob,os,collections
fs=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03_call3/pmc_w3/**/*counter_collection.csv',recursive=True)
for f in fs:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in agg.items():
        if 'wgrad' in k: print(k, {a: round(b/1e6,2) for a,b in v.items()})
EOP
