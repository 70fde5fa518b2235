"""profiles/r04_comparator.md from the JSON of `tools/microbench.py --what comparator`."""
import json
import sys

rows = [r for r in json.load(open(sys.argv[1])) if r['layer'].startswith('cmp ')]
out = ['# In-tree kernels vs the vendor libraries on the 23 ResNet-50 1x layer shapes (1024 views, bf16, one MI355X)', '',
       '`tools/microbench.py --what comparator` (median of 5 launches after 2 warm-ups, HIP events, random data): MIOpen through',
       '`torch.nn.functional.conv2d` / `aten.convolution_backward` (channels_last = NHWC, immediate mode), hipBLASLt through',
       '`torch.matmul` for the 1x1 stride-1 layers.  A yardstick only: neither library is ever on the product path.', '',
       '| layer (x count) | ours fwd / dgrad / wgrad us | MIOpen fwd / dgrad / wgrad us | hipBLASLt fwd / dgrad / wgrad us | best vendor / ours (fwd, dgrad, wgrad) | bound us (MFMA / HBM@8) |',
       '|---|---|---|---|---|---|']
tot = {'ours': [0, 0, 0], 'vendor': [0, 0, 0]}
lose = []
for r in rows:
    o, m, b = r['ours'], r['miopen'], r['hipblaslt']
    best = []
    for i in range(3):
        c = [v for v in (m[i], b[i]) if v == v]
        best.append(min(c) if c else float('nan'))
        tot['ours'][i] += r['count'] * o[i]
        tot['vendor'][i] += r['count'] * best[i]
        if best[i] < o[i]:
            lose.append((r['layer'][4:], ('fwd', 'dgrad', 'wgrad')[i], o[i], best[i]))
    f = lambda v: ' / '.join('-' if x != x else '%.0f' % x for x in v)
    out.append('| %s | %s | %s | %s | %s | %.0f / %.0f |' % (r['layer'][4:], f(o), f(m), f(b), ' / '.join('%.2f' % (best[i] / o[i]) for i in range(3)),
                                                          r['flops'] / 2.5e9, r['bytes'] / 8e6))
out += ['', 'Weighted by layer count (ms per pass over the 53 convolutions): ours fwd %.2f / dgrad %.2f / wgrad %.2f; best vendor kernel per layer fwd %.2f / dgrad %.2f / wgrad %.2f.' % (
    tuple(v / 1e3 for v in tot['ours']) + tuple(v / 1e3 for v in tot['vendor'])), '',
    'Layers where a vendor kernel is faster than the in-tree one (the targets this table sets):', '']
for l in lose:
    out.append('* %s %s: ours %.0f us, vendor %.0f us (%.0f %%)' % (l[0], l[1], l[2], l[3], 100 * (l[2] / l[3] - 1)))
if not lose:
    out.append('* none')
print('\n'.join(out))
