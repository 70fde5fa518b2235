"""Per-layer roofline table from tools/microbench.py's JSON (profiles/rNN_microbench.json):
for every distinct ResNet-50 conv shape the time max(FLOPs / MFMA peak, algorithmic bytes / HBM peak) would allow,
the measured fwd / dgrad / wgrad times and their ratio, and the per-step totals ("measured / per-layer bound").

python tools/per_layer_roofline.py profiles/r01_microbench.json > profiles/r01_per_layer_roofline.md
"""
import json
import re
import sys

MFMA = 2.5e15        # bf16 dense FLOP/s
HBM = 8.0e12         # spec B/s (6.29e12 measured float4 copy)
V = 1024


def main():
    # optional: --elt 4 --terms 6,3,3 = fp32 storage with split-bf16 arithmetic (the parity mode): bytes at 4 B / element and the
    # MFMA bound per pass at 2.5 PFLOP/s / terms (forward, dgrad, wgrad)
    elt, terms = 2.0, (1, 1, 1)
    if '--elt' in sys.argv:
        elt = float(sys.argv[sys.argv.index('--elt') + 1])
    if '--terms' in sys.argv:
        terms = tuple(int(t) for t in sys.argv[sys.argv.index('--terms') + 1].split(','))
    rows = json.load(open(sys.argv[1]))
    print('| layer (x count) | GFLOP | alg. MB (fwd) | bound us (MFMA / HBM@8 / HBM@6.29) | fwd us (frac) | dgrad us (frac) | wgrad us (frac) |')
    print('|---|---|---|---|---|---|---|')
    tot = dict(bound=0.0, bound629=0.0, fwd=0.0, dgrad=0.0, wgrad=0.0)
    for r in rows:
        m = re.match(r'(\d+)x\d+ (\d+)->(\d+) k(\d) s(\d) x(\d+)', r['layer'])
        if not m or 'fwd_us' not in r:
            continue
        H, ci, co, k, s, cnt = map(int, m.groups())
        OH = H // s
        in_px = H * H if (k == 3 or s == 1) else OH * OH          # a strided 1x1 touches every s-th pixel only
        by = elt * V * (in_px * ci + OH * OH * co) + elt * k * k * ci * co
        fl = r['flops']
        t_m, t_h, t_h2 = fl * terms[0] / MFMA * 1e6, by / HBM * 1e6, by / 6.29e12 * 1e6
        b, b2 = max(t_m, t_h), max(t_m, t_h2)
        bd, bw = max(fl * terms[1] / MFMA * 1e6, t_h), max(fl * terms[2] / MFMA * 1e6, t_h)      # = b when terms are (1, 1, 1)
        print('| %s | %.0f | %.0f | %.0f (%.0f / %.0f / %.0f) | %.0f (%.2f) | %.0f (%.2f) | %.0f (%.2f) |' % (
            r['layer'], fl / 1e9, by / 1e6, b, t_m, t_h, t_h2, r['fwd_us'], b / r['fwd_us'], r['dgrad_us'], bd / r['dgrad_us'],
            r['wgrad_us'], bw / r['wgrad_us']))
        tot['bound'] += cnt * b; tot['bound629'] += cnt * b2
        for key in ('fwd', 'dgrad', 'wgrad'):
            tot[key] += cnt * r[key + '_us']
    print()
    print('Per step (53 convs after the stem, 1024 views): bound %.2f ms per pass at 8 TB/s (%.2f ms at 6.29 TB/s); measured fwd %.2f ms '
          '(%.0f %% of the bound), dgrad %.2f ms (%.0f %%; without the fused BN-backward epilogue), wgrad %.2f ms (%.0f %%).' % (
              tot['bound'] / 1e3, tot['bound629'] / 1e3, tot['fwd'] / 1e3, 100 * tot['bound'] / tot['fwd'],
              tot['dgrad'] / 1e3, 100 * tot['bound'] / tot['dgrad'], tot['wgrad'] / 1e3, 100 * tot['bound'] / tot['wgrad']))
    print('"frac" = bound / measured (1.00 = on the roofline).  The bound ignores the BatchNorm / ReLU passes between the '
          'convolutions, which cost a further ~23 ms per step of pure streaming (5.4-5.9 TB/s measured).')


if __name__ == '__main__':
    main()
