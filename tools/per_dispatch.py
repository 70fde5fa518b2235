"""Per-dispatch view of one training step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

python tools/per_dispatch.py gpurun_out/r03_final/stats_kernel_trace.csv.gz > profiles/r03_per_dispatch.md

The LAST complete step of the run is cut out (a step starts at the `pack_views` dispatch) and printed in launch order:
index, duration, grid, kernel (template arguments kept, namespaces dropped) -- the in-step counterpart of
tools/microbench.py's stand-alone layer times -- followed by totals per kernel family."""
import csv
import gzip
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('unsigned short', 'bf16')


def main():
    path = sys.argv[1]
    op = gzip.open if path.endswith('.gz') else open
    rows = []
    with op(path, 'rt') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
                         r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith('pack_views')]
    if len(starts) < 2:
        sys.exit('need at least two steps in the trace')
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    wall = (step[-1][1] - step[0][0]) / 1e6
    busy = sum(r[1] - r[0] for r in step) / 1e6
    print('# One training step, dispatch by dispatch (rocprofv3 kernel trace, ResNet-50 1x, 224 px, 512 images, bf16)\n')
    print('%d dispatches, first start to last end %.2f ms, summed kernel time %.2f ms.\n' % (len(step), wall, busy))
    print('| # | us | grid x wg | kernel |')
    print('|---|---|---|---|')
    fam = defaultdict(lambda: [0, 0.0])
    for i, (s, e, n, g, w) in enumerate(step):
        us = (e - s) / 1e3
        f = re.sub(r'<.*$', '', n)
        fam[f][0] += 1
        fam[f][1] += us
        if us >= 40.0:
            print('| %d | %.0f | %s x %s | `%s` |' % (i, us, g, w, n[:150]))
    print('\n(dispatches shorter than 40 us are listed in the family totals only)\n')
    print('| kernel family | dispatches | ms |')
    print('|---|---|---|')
    for f, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print('| `%s` | %d | %.3f |' % (f, c, us / 1e3))


if __name__ == '__main__':
    main()
