#!/bin/bash
# round 6, call 52: Gram launches of the fused tail on six bf16-piece terms instead of exact fp32 MFMA (SIMCLR_GRAM_SPLIT=0 = exact) -- tests, parity, A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call52
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gram or fused_bn_apply or bn_apply_tail or fold or batch32 or reference_source_fixtures or parity_at_baseline or resnet50 or determinis" > "$OUT/pytest.txt" 2>&1; tail -6 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_f32"
for rep in 1 2 3; do
  env SIMCLR_GRAM_SPLIT=0 timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_old_$rep.json" 2>> "$OUT/err.txt"
  timeout 300 $B --steps 8 --warmup 3 --prof_steps 2 > "$OUT/bench_new_$rep.json" 2>> "$OUT/err.txt"
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        m = d['parity']['modes']['value (f32/f16x3_3)']
        print(os.path.basename(f), d['ms_per_step'], d['kernels'].get('conv_wgrad', {}).get('ms_per_step'), 'loss_rel %.3e emb_abs %.3e met %s' % (m['loss_rel'], m['emb_abs'], m['north_star_met']), {k: (v['loss_rel'], v['emb_abs']) for k, v in m['cases'].items()})
    except Exception as e:
        print(os.path.basename(f), 'failed', e)
PY
tail -3 "$OUT/err.txt"
