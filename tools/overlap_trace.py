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
    is_coll = lambda n: ('nccl' in n.lower()) or ('rccl' in n.lower())
    skip = lambda n: 'aug_' in n or 'at::native' in n or 'FillFunctor' in n
    coll = [r for r in rows if is_coll(r[2])]
    comp = [r for r in rows if not is_coll(r[2]) and not skip(r[2])]
    # union of compute intervals
    merged = []
    for s, e, *_ in comp:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [m[0] for m in merged]

    def overlap(s, e):
        i = max(0, bisect.bisect_right(starts, s) - 1)
        tot = 0
        while i < len(merged) and merged[i][0] < e:
            tot += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        return tot

    def kind(n):
        for k in ('AllGather', 'ReduceScatter', 'AllReduce', 'Broadcast', 'SendRecv'):
            if k.lower() in n.lower():
                return k
        return 'other'
    by = {}
    for s, e, n, q, st in coll:
        d = by.setdefault(kind(n), dict(launches=0, ns=0, hidden_ns=0, queues=set(), streams=set(), longest_us=0.0))
        d['launches'] += 1
        d['ns'] += e - s
        d['hidden_ns'] += overlap(s, e)
        d['queues'].add(q); d['streams'].add(st)
        d['longest_us'] = max(d['longest_us'], (e - s) / 1e3)
    span = (rows[-1][1] - rows[0][0]) if rows else 0
    out = dict(source='rocprofv3 --kernel-trace of bench.py under SIMCLR_FORCE_COLLECTIVES=1 (one rank, every collective issued)',
               steps=args.steps, compute_kernels=len(comp), collective_kernels=len(coll),
               compute_queues=sorted({r[3] for r in comp}), collective_queues=sorted({r[3] for r in coll}),
               compute_busy_ms_per_step=round(sum(m[1] - m[0] for m in merged) / 1e6 / args.steps, 3),
               trace_span_ms=round(span / 1e6, 3), collectives={})
    for k, d in sorted(by.items()):
        out['collectives'][k] = dict(launches_per_step=round(d['launches'] / args.steps, 2), us_per_step=round(d['ns'] / 1e3 / args.steps, 1),
                                     hidden_behind_compute_frac=round(d['hidden_ns'] / max(d['ns'], 1), 4),
                                     exposed_us_per_step=round((d['ns'] - d['hidden_ns']) / 1e3 / args.steps, 1),
                                     longest_kernel_us=round(d['longest_us'], 1), queues=sorted(d['queues']), streams=sorted(d['streams']))
    print(json.dumps(out, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
