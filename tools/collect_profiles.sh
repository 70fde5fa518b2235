#!/bin/bash
# Copies the judged summaries of one `tools/gpu_round_checks.sh <tag> full` run into profiles/ (tracked):
#   bash tools/collect_profiles.sh r03 gpurun_out/r03_final2
set -eu
RN=$1; D=$2; P=profiles
python - "$D" "$P/${RN}_bench_line.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1])
json.dump(d, open(sys.argv[2], 'w'), indent=1)
EOP
for c in cfg4 cfg5_b256; do
  python - "$D/bench_$c.json" "$P/${RN}_bench_$c.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
json.dump(d, open(sys.argv[2], 'w'), indent=1)
EOP
done
cp "$D/stats_kernel_stats.csv" "$P/${RN}_bench_kernel_stats.csv"
cp "$D/stats_domain_stats.csv" "$P/${RN}_bench_domain_stats.csv"
cp "$D/stats_kernel_trace.csv.gz" "$P/${RN}_bench_kernel_trace.csv.gz"
cp "$D"/pmc_f/*counter_collection.csv.gz "$P/${RN}_pmc_fetch_counter_collection.csv.gz"
cp "$D"/pmc_w/*counter_collection.csv.gz "$P/${RN}_pmc_write_counter_collection.csv.gz"
python tools/pmc_traffic.py "$P/${RN}_pmc_fetch_counter_collection.csv.gz" "$P/${RN}_pmc_write_counter_collection.csv.gz" "$P/${RN}_pmc_traffic.json"
python tools/pmc_traffic.py --ntxent "$D"/pmc_nt_f/*counter_collection.csv "$D"/pmc_nt_w/*counter_collection.csv "$D"/pmc_nt_f/*kernel_trace.csv "$P/${RN}_pmc_traffic.json"
cp "$D/microbench.json" "$P/${RN}_microbench.json"
grep -v amdgpu.ids "$D/microbench.txt" > "$P/${RN}_microbench_per_layer.txt"
python tools/per_layer_roofline.py "$P/${RN}_microbench.json" > "$P/${RN}_per_layer_roofline.md"
python tools/per_dispatch.py "$P/${RN}_bench_kernel_trace.csv.gz" > "$P/${RN}_per_dispatch.md"
ls -la $P/${RN}_*
