#!/bin/bash
# round 5, call 5: the corrected last test, then the final bench line (split-bf16 stem in the parity mode), the parity-mode kernel
# stats + PMC passes of that build, and the cfg5 line (ResNet-152 3x + SK at its real per-GPU share)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_call5
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "resnet152 or stem_conv" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log" | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
T0=$SECONDS
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-260 "$OUT/bench.json" | tail -1; echo "bench: $((SECONDS - T0)) s"
timeout 700 python bench.py --resnet_depth 152 --width_multiplier 3 --sk_ratio 0.0625 --per_gpu_batch 256 --steps 4 --warmup 2 --no_cpu_baseline --no_f32 --no_pmc --no_parity --prof_steps 1 > "$OUT/bench_cfg5_b256.json" 2> "$OUT/bench_cfg5_b256.err"; cut -c1-260 "$OUT/bench_cfg5_b256.json" | tail -1
cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --no_cpu_baseline --no_kernel_events --no_f32 --no_pmc --no_parity --dtype f32 --f32_matmul bf16x6_3"
mkdir -p "$OUT/parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/parity" -o stats -- $P --steps 3 --warmup 1 > "$OUT/parity/prof.log" 2>&1
gzip -f "$OUT"/parity/*kernel_trace.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/parity/pmc_f" -o f -- $P --steps 1 --warmup 1 > "$OUT/parity/pmc_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/parity/pmc_w" -o w -- $P --steps 1 --warmup 1 > "$OUT/parity/pmc_w.log" 2>&1
gzip -f "$OUT"/parity/pmc_f/*counter_collection.csv "$OUT"/parity/pmc_w/*counter_collection.csv 2>/dev/null
rm -f "$OUT"/parity/*agent_info.csv
cd "$R"; ls "$OUT" "$OUT/parity"; echo "total: $((SECONDS - T0)) s"
