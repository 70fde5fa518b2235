"""HBM traffic per kernel from two rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json).

On the GPU box (separate passes, as MI355X_MICROARCH.md prescribes -- never combined with trace domains):
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events
then
  python tools/pmc_traffic.py gpurun_out/pmc_f/*counter_collection.csv gpurun_out/pmc_w/*counter_collection.csv profiles/r02_pmc_traffic.json

Counters are in KB.  FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by exactly 2x on gfx950 (guide, HBM
section); every streaming load in this library is 16 B/lane, so fetch_corrected = 2 x raw.  WRITE_SIZE is taken as is.
The run has 2 steps (1 warm-up + 1 timed): per-step values are the totals halved.
"""
import csv
import gzip
import json
import sys
from collections import defaultdict

STEPS = 2.0


def load(path, counter):
    op = gzip.open if path.endswith('.gz') else open
    tot, calls = defaultdict(float), defaultdict(int)
    with op(path, 'rt') as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            tot[r['Kernel_Name']] += float(r['Counter_Value']) * 1024.0
            calls[r['Kernel_Name']] += 1
    return tot, calls


def ntxent_rows(fpath, wpath, tpath):
    """HBM traffic of the NT-Xent / l2norm kernels from PMC passes of `tools/microbench.py --what ntxent` (n = N = 512,
    n 512 / N 4096 (cfg3 per-GPU shape), n 256 / N 2048, n = N = 4096) with per-dispatch durations from the kernel trace."""
    fetch, calls = load(fpath, 'FETCH_SIZE')
    write, _ = load(wpath, 'WRITE_SIZE')
    dur = defaultdict(float)
    op = gzip.open if tpath.endswith('.gz') else open
    with op(tpath, 'rt') as f:
        for r in csv.DictReader(f):
            dur[r['Kernel_Name']] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    rows = []
    for n in sorted(set(fetch) | set(write)):
        if 'ntxent' not in n and 'l2norm' not in n:
            continue
        c = max(calls.get(n, 0), 1)
        by = 2 * fetch.get(n, 0) + write.get(n, 0)
        rows.append(dict(kernel=n.replace('void (anonymous namespace)::', '')[:70], dispatches=c,
                         fetch_corrected_bytes_per_dispatch=2 * fetch.get(n, 0) / c, write_bytes_per_dispatch=write.get(n, 0) / c,
                         avg_us=dur.get(n, 0.0) / c / 1e3, hbm_gbps=(by / dur[n]) if dur.get(n) else None))
    return rows


def main():
    if sys.argv[1] == '--ntxent':
        fpath, wpath, tpath, out = sys.argv[2:6]
        res = json.load(open(out))
        res['ntxent'] = dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/microbench.py --what ntxent: all four '
                                  'shapes pooled per kernel; FETCH x2; durations from the FETCH pass kernel trace.  The kernels are bound by '
                                  'the fp32 matrix pipe and launch latency, not by HBM: GB/s far below the roof is expected.',
                             per_kernel=ntxent_rows(fpath, wpath, tpath))
        json.dump(res, open(out, 'w'), indent=1)
        for r in res['ntxent']['per_kernel']:
            print('%-70s %4d x  fetch %8.3f MB  write %8.3f MB  %7.1f us  %s GB/s' % (
                r['kernel'], r['dispatches'], r['fetch_corrected_bytes_per_dispatch'] / 1e6, r['write_bytes_per_dispatch'] / 1e6,
                r['avg_us'], None if r['hbm_gbps'] is None else round(r['hbm_gbps'], 1)))
        return
    fpath, wpath, out = sys.argv[1:4]
    fetch, calls = load(fpath, 'FETCH_SIZE')
    write, _ = load(wpath, 'WRITE_SIZE')
    names = sorted(set(fetch) | set(write), key=lambda n: -(2 * fetch.get(n, 0) + write.get(n, 0)))
    per = []
    for n in names:
        per.append(dict(kernel=n.replace('void (anonymous namespace)::', '')[:110], calls_per_step=calls.get(n, 0) / STEPS,
                        fetch_raw_bytes=fetch.get(n, 0) / STEPS, write_bytes=write.get(n, 0) / STEPS))
    fam = [n for n in names if 'conv_igemm' in n]          # persistent + the plain fallback: the family bench.py calls conv_igemm
    launches = sum(calls[n] for n in fam) / STEPS
    fr = sum(fetch.get(n, 0) for n in fam) / STEPS
    wr = sum(write.get(n, 0) for n in fam) / STEPS
    # bench.py also times the two-view augmentation once per run, outside the training step: not part of the step's traffic
    side = [n for n in names if 'aug_' in n]
    total = sum(2 * fetch.get(n, 0) + write.get(n, 0) for n in names if n not in side) / STEPS
    side_total = sum(2 * fetch.get(n, 0) + write.get(n, 0) for n in side)
    res = dict(
        note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 1`; KB counters '
             'x1024, halved per step; FETCH_SIZE x2 for 16 B/lane loads (MI355X_MICROARCH.md), WRITE_SIZE as is.',
        kernel='conv_igemm', kernel_family='conv_igemm_persistent / conv_igemm (fwd + dgrad)', kernel_dispatches_per_step=launches,
        family_fetch_corrected_bytes_per_step=2 * fr, family_write_bytes_per_step=wr, family_bytes_per_step=2 * fr + wr,
        fetch_raw_bytes_per_launch=fr / max(launches, 1), fetch_corrected_bytes_per_launch=2 * fr / max(launches, 1),
        write_bytes_per_launch=wr / max(launches, 1), traffic_bytes_per_launch=(2 * fr + wr) / max(launches, 1),
        step_total_bytes=total, side_measurement_bytes_excluded=side_total, per_kernel=per[:40])
    json.dump(res, open(out, 'w'), indent=1)
    print('step total %.1f GB; conv_igemm family %.3f GB/launch over %d launches/step' % (
        total / 1e9, res['traffic_bytes_per_launch'] / 1e9, launches))
    for p in per[:14]:
        print('%-100s %5.0f calls  fetch(x2) %7.2f GB  write %7.2f GB' % (p['kernel'][:100], p['calls_per_step'],
                                                                          2 * p['fetch_raw_bytes'] / 1e9, p['write_bytes'] / 1e9))


if __name__ == '__main__':
    main()
