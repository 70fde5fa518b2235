#!/bin/bash
# round 6, call 34: full GPU suite on the build with the fp16 fma-mix split; clocks / power sampled during a headline-mode run and a speed-mode run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_call34
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
B="python $R/bench.py --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events"
sample() {  # $1 = output file; samples until the file $1.stop exists
  while [ ! -f "$1.stop" ]; do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ';' >> "$1"; echo >> "$1"
    sleep 0.5
  done
}
rocm-smi --showclocks --showpower > "$OUT/smi_idle.txt" 2>&1
for mode in f32 bf16; do
  rm -f "$OUT/smi_$mode.txt.stop"; sample "$OUT/smi_$mode.txt" & SP=$!
  timeout 300 $B --dtype $mode --steps 60 --warmup 10 > "$OUT/bench_$mode.json" 2>> "$OUT/err.txt"
  touch "$OUT/smi_$mode.txt.stop"; wait $SP; rm -f "$OUT/smi_$mode.txt.stop"
done
python - <<PY
import re, json
for mode in ('f32', 'bf16'):
    try:
        d = json.loads(open('$OUT/bench_%s.json' % mode).read().strip().splitlines()[-1]); print(mode, d['ms_per_step'])
    except Exception as e: print(mode, 'failed', e)
    lines = open('$OUT/smi_%s.txt' % mode).read().splitlines()
    sc = [int(m.group(1)) for l in lines for m in [re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', l)] if m]
    pw = [float(m.group(1)) for l in lines for m in [re.search(r'Power \(W\): ([\d.]+)', l)] if m]
    print(mode, 'samples', len(lines), 'sclk', sorted(sc)[:3], sorted(sc)[-3:], 'power', sorted(pw)[:2], sorted(pw)[-3:])
PY
head -3 "$OUT/smi_f32.txt"
