#!/bin/bash
# One gpurun call that produces everything a round needs (run ON the GPU box, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round_checks.sh rNN [full|quick]'
# quick: conv/BN parity subset + bench (about 1.5 min);  full: whole GPU suite, smoke, bench with CPU baseline and fp32 mode,
# microbench, rocprofv3 kernel stats and the two PMC passes (about 12 min).  Everything lands in gpurun_out/<tag>/;
# copy what should be judged into profiles/ afterwards (tools/pmc_traffic.py, tools/per_layer_roofline.py).
set -u
TAG=${1:-r00}
MODE=${2:-quick}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
if [ "$MODE" = "quick" ]; then
  timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "test_conv_fwd or dgrad or batch_norm or pooling" 2>&1 | tail -3
  timeout 120 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_f32 --no_pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
  python -c "import json,sys;d=json.load(open('$OUT/bench_quick.json'));print(d['value'],d['ms_per_step']);[print(k,v) for k,v in d['kernels'].items()]"
  exit 0
fi
if [ "$MODE" != "prof" ]; then      # prof: bench line, microbench, rocprofv3 stats / trace and the two PMC passes only
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"; grep -n "^FAILED" "$OUT/pytest_gpu.log" | head
fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json" | tail -1
timeout 300 python tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.txt" 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B --steps 3 --warmup 1 > "$OUT/prof.log" 2>&1
gzip -f "$OUT"/*kernel_trace.csv 2>/dev/null      # per-dispatch durations (tools/per_dispatch.py)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 1 --warmup 1 > "$OUT/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 1 --warmup 1 > "$OUT/pmc_w.log" 2>&1
gzip -f "$OUT"/pmc_f/*counter_collection.csv "$OUT"/pmc_w/*counter_collection.csv 2>/dev/null
if [ "$MODE" = "prof" ]; then ls "$OUT"; exit 0; fi
# north_star: rocprof-reported HBM traffic of the NT-Xent kernels (cfg2 and cfg3 shapes: tools/microbench.py --what ntxent)
NT="python $R/tools/microbench.py --what ntxent --iters 3 --out $OUT/mb_ntxent.json"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_nt_f" -o f -- $NT > "$OUT/pmc_nt_f.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_nt_w" -o w -- $NT > "$OUT/pmc_nt_w.log" 2>&1
cd "$R"
if [ "$MODE" = "full" ]; then
  # BASELINE configs[3] and [4] on one GPU (per-GPU share of the 8-GPU global batch): value, step_mfma_frac, peak HBM
  timeout 400 python bench.py --resnet_depth 50 --width_multiplier 2 --sk_ratio 0.0625 --steps 8 --warmup 3 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-260 "$OUT/bench_cfg4.json" | tail -1
  # cfg5 at its real per-GPU share (global 2048 / 8 = 256 images): peaks at 192 GB of the 288 GB (104 GB at 128 images, r03_call5)
  timeout 700 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --no_pmc --prof_steps 1 > "$OUT/bench_cfg5_b256.json" 2> "$OUT/bench_cfg5_b256.err"; cut -c1-260 "$OUT/bench_cfg5_b256.json" | tail -1
fi
if [ "$MODE" = "full" ]; then
  # collective C on one GPU: two ranks over gloo sharing cuda:0, with and without the peer-mapped statistics exchange
  SIMCLR_PEER_STATS=1 timeout 400 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --per_gpu_batch 64 --no_cpu_baseline --no_f32 --no_pmc --no_kernel_events > "$OUT/bench_2rank_peer.json" 2> "$OUT/bench_2rank_peer.err"; cut -c1-200 "$OUT/bench_2rank_peer.json" | tail -1
fi
ls "$OUT"
