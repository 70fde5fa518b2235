#!/usr/bin/env python
"""(layer class -> kernel instantiation) table of the forward / data-gradient convolution launches.

    python tools/instantiation_table.py [--views 1024] [--out profiles/r06_instantiations.txt]

Runs WITHOUT a GPU: with SIMCLR_DRY_RUN=1 the convolution entry points of libsimclr_hip.so take every launch decision
(launch_igemm_one: tile shape, halo window, wide eight-phase tile, split tail, pre-split weights, compile-time epilogue
specialisations) and record it instead of launching (simclr_conv2d_last_instantiation).  One row per (ResNet-50 1x layer class at
224 px) x (kind of launch a training step issues for it) x (storage / matrix arithmetic).  tests/test_abi.py regenerates the
table and compares it with the committed file, so a change of the selection rules is a visible diff, not a silent one.
"""
import argparse
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

# ResNet-50 1x at 224 px (tf2/resnet.py:385-487, 709-729): (name, H of the INPUT, Cin, Cout, k, stride, count per step)
LAYERS = [
    ('g1 reduce 1x1 64->64 (first block)', 56, 64, 64, 1, 1, 1),
    ('g1 reduce 1x1 256->64', 56, 256, 64, 1, 1, 2),
    ('g1 3x3 64->64', 56, 64, 64, 3, 1, 3),
    ('g1 expand 1x1 64->256', 56, 64, 256, 1, 1, 3),
    ('g1 shortcut 1x1 64->256', 56, 64, 256, 1, 1, 1),
    ('g2 reduce 1x1 256->128 @56', 56, 256, 128, 1, 1, 1),
    ('g2 3x3 s2 128->128 56->28', 56, 128, 128, 3, 2, 1),
    ('g2 shortcut 1x1 s2 256->512', 56, 256, 512, 1, 2, 1),
    ('g2 reduce 1x1 512->128', 28, 512, 128, 1, 1, 3),
    ('g2 3x3 128->128', 28, 128, 128, 3, 1, 3),
    ('g2 expand 1x1 128->512', 28, 128, 512, 1, 1, 4),
    ('g3 reduce 1x1 512->256 @28', 28, 512, 256, 1, 1, 1),
    ('g3 3x3 s2 256->256 28->14', 28, 256, 256, 3, 2, 1),
    ('g3 shortcut 1x1 s2 512->1024', 28, 512, 1024, 1, 2, 1),
    ('g3 reduce 1x1 1024->256', 14, 1024, 256, 1, 1, 5),
    ('g3 3x3 256->256', 14, 256, 256, 3, 1, 5),
    ('g3 expand 1x1 256->1024', 14, 256, 1024, 1, 1, 6),
    ('g4 reduce 1x1 1024->512 @14', 14, 1024, 512, 1, 1, 1),
    ('g4 3x3 s2 512->512 14->7', 14, 512, 512, 3, 2, 1),
    ('g4 shortcut 1x1 s2 1024->2048', 14, 1024, 2048, 1, 2, 1),
    ('g4 reduce 1x1 2048->512', 7, 2048, 512, 1, 1, 2),
    ('g4 3x3 512->512', 7, 512, 512, 3, 1, 2),
    ('g4 expand 1x1 512->2048', 7, 512, 2048, 1, 1, 3),
    ('head dense 2048->2048', 1, 2048, 2048, 1, 1, 2),
    ('head dense 2048->128', 1, 2048, 128, 1, 1, 1),
]
FAKE = ctypes.c_void_p(1 << 20)          # never dereferenced in a dry run
MODES = [('bf16', 1, (0, 0)), ('f32 exact', 0, (0, 0)), ('f32 bf16x6_3', 0, (6, 3)), ('f32 f16x3_3', 0, (13, 3))]


def rows(views):
    os.environ['SIMCLR_DRY_RUN'] = '1'
    from simclr_amd._lib import lib
    L = lib()
    out = []
    last = lambda: L.conv2d_last_instantiation().decode()
    try:
        for mode, dtype, terms in MODES:
            L.set_f32_matmul(*terms)
            for name, H, Cin, Cout, k, s, cnt in LAYERS:
                V = views
                pad = (k - 1) // 2
                OH = (H + (k - 1) - k) // s + 1
                M = V * OH * OH
                nslot = L.conv2d_stats_slots(M, Cout)
                kinds = []
                # forward with BatchNorm statistics in the epilogue (every convolution feeds a BatchNorm, tf2/resnet.py:183-208 + 31-78)
                if dtype == 1:
                    L.conv2d_fwd(FAKE, FAKE, FAKE, FAKE, nslot, V, H, H, Cin, OH, OH, Cout, k, k, s, pad, dtype, None)
                else:
                    L.conv2d_fwd_pivoted(FAKE, FAKE, FAKE, FAKE, nslot, FAKE, V, H, H, Cin, OH, OH, Cout, k, k, s, pad, dtype, None)
                kinds.append(('fwd + statistics', last()))
                if dtype == 1 and 'expand' in name:      # fused bottleneck tail: conv3 + bn3 + shortcut + ReLU + mask bits (bf16)
                    L.conv2d_fwd_bn_apply(FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, None, None, 1, FAKE, V, H, H, Cin, OH, OH, Cout, k, k, s, pad, dtype, None)
                    kinds.append(('fwd + fused BatchNorm apply / residual / ReLU', last()))
                if H > 1 or True:
                    L.conv2d_dgrad(FAKE, FAKE, FAKE, 0, V, H, H, Cin, OH, OH, Cout, k, k, s, pad, dtype, None)
                    kinds.append(('dgrad', last()))
                if s == 1 and H > 1:
                    nsl_in = L.conv2d_stats_slots(V * H * H, Cin)
                    # dgrad + BatchNorm-backward reduce of the producer: mode 2 (mask recomputed from the BatchNorm input), store
                    L.conv2d_dgrad_bn(FAKE, FAKE, FAKE, 0, FAKE, None, FAKE, FAKE, FAKE, FAKE, 2, FAKE, nsl_in, V, H, H, Cin, OH, OH, Cout,
                                      k, k, 1, pad, dtype, None)
                    kinds.append(('dgrad + BN-backward reduce (mode 2)', last()))
                    if 'reduce' in name:                 # conv1 of a block: accumulates into the shortcut gradient, mask bits, sums only
                        L.conv2d_dgrad_bn(FAKE, FAKE, FAKE, 1, None, FAKE, None, None, None, None, 4, FAKE, nsl_in, V, H, H, Cin, OH, OH,
                                          Cout, k, k, 1, pad, dtype, None)
                        kinds.append(('dgrad + accumulate + BN-backward sums (mode 4)', last()))
                for kind, inst in kinds:
                    out.append((mode, name, cnt, kind, inst))
    finally:
        L.set_f32_matmul(0, 0)
        os.environ.pop('SIMCLR_DRY_RUN', None)
    return out


def render(views):
    lines = ['# forward / data-gradient kernel instantiation per (ResNet-50 1x layer class, launch kind, storage mode) at %d views' % views,
             '# generated by tools/instantiation_table.py (SIMCLR_DRY_RUN=1: decisions only, no device); columns:',
             '# mode | layer class | launches per step of this class | launch kind | instantiation record', '']
    for mode, name, cnt, kind, inst in rows(views):
        lines.append(' | '.join([mode, name, 'x%d' % cnt, kind, inst]))
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=1024)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    txt = render(a.views)
    if a.out:
        open(a.out, 'w').write(txt)
        print('wrote %d rows to %s' % (txt.count('\n') - 4, a.out))
    else:
        sys.stdout.write(txt)
