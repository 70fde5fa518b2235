#!/bin/bash
# round 4, call 6: peer-mapped SyncBN statistics exchange (2 / 4 processes on one GPU), wgrad side stream A/B, bench line with in-run PMC
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call6
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "peer" > "$OUT/pytest_peer.log" 2>&1
tail -3 "$OUT/pytest_peer.log" | cut -c1-300; grep -n "^FAILED\|^E  \|Error" "$OUT/pytest_peer.log" | head -20 | cut -c1-250
B="python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_f32 --no_pmc"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
run default_a X=1
run sidestream_a SIMCLR_WGRAD_STREAM=1
run default_b X=1
run sidestream_b SIMCLR_WGRAD_STREAM=1
timeout 500 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
python - <<'EOP'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_call6/bench*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-28s %8.3f ms %8.1f img/s' % (os.path.basename(f), d['ms_per_step'], d['value']), (d.get('roofline') or {}).get('frac'))
        if 'full' in f:
            r=d['roofline']; print({k:r.get(k) for k in ('traffic','traffic_measured_in_run','traffic_over_algorithmic','step_total_traffic_gb','frac')}); print('f32', d['f32_mode']); print('parity', d['parity_mode'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
EOP
