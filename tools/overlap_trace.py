"""Overlap of the collective-library kernels with compute, from a rocprofv3 kernel trace of a one-rank step with every collective
issued:

    cd /tmp && SIMCLR_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace --output-format csv -d OUT -o fc -- \
        python /root/repo/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_parity --no_f32 --no_kernel_events
    python tools/overlap_trace.py OUT --steps 5 --out profiles/r06_overlap_forced_collectives.json

With SIMCLR_FORCE_COLLECTIVES=1 a single rank builds the three communicators (hidden / statistics / gradients) and issues collective A
(all-gather + reduce-scatter of the hidden block, tf2/objective.py:92-127), B (bucketed gradient all-reduce, tf2/run.py:614-622) and C
(SyncBatchNormalization moments, tf2/resnet.py:50-60) -- each is the identity on one rank, but the RCCL kernels, their streams and the
event dependencies are the ones an 8-GPU job runs.  Reported per collective kind: launches per step, kernel time, and the part of that
time during which a COMPUTE kernel of this library was executing on another queue (= hidden behind compute).  What one GPU cannot show
is wire time: the projection in BASELINE.md multiplies message sizes by the guide's xGMI figures, labelled as such."""
import argparse
import csv
import glob
import json
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--steps', type=int, default=5, help='steps the traced command ran (warm-up + timed)')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit('no kernel trace under %s' % args.dir)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    skip = lambda n: 'aug_' in n
    rows = [r for r in rows if not skip(r[2])]
    # streams: the launch stream carries the compute kernels; every other stream with recurring work is a communicator's stream.
    # On ONE rank RCCL executes a collective as a device-to-device copy (__amd_rocclr_copyBuffer) or nothing at all (in-place), and the
    # SyncBN moments go through csrc/comm.hip's stats_exchange / the library: what the trace shows is therefore the STREAM STRUCTURE --
    # which work runs beside the compute stream and how much of it is covered by compute kernels -- not ring kernels.
    from collections import Counter, defaultdict
    per_stream = Counter((r[3], r[4]) for r in rows)
    main = per_stream.most_common(1)[0][0]
    comp = [r for r in rows if (r[3], r[4]) == main]
    side = defaultdict(list)
    for r in rows:
        if (r[3], r[4]) != main:
            side[(r[3], r[4])].append(r)
    merged = []
    for s_, e_, *_ in comp:
        if merged and s_ <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e_)
        else:
            merged.append([s_, e_])
    import bisect
    starts = [m[0] for m in merged]

    def overlap(s_, e_):
        i = max(0, bisect.bisect_right(starts, s_) - 1)
        tot = 0
        while i < len(merged) and merged[i][0] < e_:
            tot += max(0, min(e_, merged[i][1]) - max(s_, merged[i][0]))
            i += 1
        return tot

    span = (rows[-1][1] - rows[0][0]) if rows else 0
    out = dict(source='rocprofv3 --kernel-trace of bench.py under SIMCLR_FORCE_COLLECTIVES=1 (one rank, every collective issued)',
               note='one rank: RCCL runs each collective as a device copy on its communicator stream (no ring kernels); the table shows the '
                    'stream structure and how much of the side-stream work is covered by compute kernels of the launch stream',
               steps=args.steps, launch_stream=dict(queue=main[0], stream=main[1], kernels=len(comp),
                                                    busy_ms_per_step=round(sum(m[1] - m[0] for m in merged) / 1e6 / args.steps, 3)),
               trace_span_ms=round(span / 1e6, 3), side_streams={})
    t_first = comp[0][0] if comp else 0
    init = {}
    for key, rs in sorted(side.items(), key=lambda kv: -len(kv[1])):
        if max(r[1] for r in rs) <= t_first:          # finished before the first compute kernel: communicator set-up (ncclCommInit)
            init['queue %s / stream %s' % key] = dict(kernels=len(rs), kinds=dict(Counter(r[2].split('(')[0][:40] for r in rs).most_common(3)))
            continue
        if len(rs) < args.steps:          # one-off work
            continue
        ns = sum(r[1] - r[0] for r in rs)
        hid = sum(overlap(r[0], r[1]) for r in rs)
        names = Counter(r[2].split('(')[0][:60] for r in rs)
        out['side_streams']['queue %s / stream %s' % key] = dict(
            launches_per_step=round(len(rs) / args.steps, 1), us_per_step=round(ns / 1e3 / args.steps, 1),
            covered_by_compute_frac=round(hid / max(ns, 1), 4), exposed_us_per_step=round((ns - hid) / 1e3 / args.steps, 1),
            longest_us=round(max(r[1] - r[0] for r in rs) / 1e3, 1), kernels=dict(names.most_common(4)))
    out['set_up_streams'] = init
    print(json.dumps(out, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
