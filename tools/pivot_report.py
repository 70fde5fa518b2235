"""Pivoted BatchNorm statistics of the fp32 mode (simclr_conv2d_fwd_pivoted) vs raw fp32 moments (SIMCLR_BN_PIVOT=0):
    python tools/pivot_report.py [--r152] [--randbn] [--out gpurun_out/pivot_report.json]
kernel-level errors on outputs with |mean| >> sigma, and (optionally) the two step cases VERDICT r03 item 8 names: ResNet-152 3x +
SK on i.i.d.-noise inputs, and the ResNet-50 / 224 px fixed-gate step with randomised BatchNorm parameters.  Measurement tool:
prints every gate, asserts nothing."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests import gpu_checks as gc  # noqa: E402


def show(tag, res, table):
    for r in res:
        flag = 'ok  ' if r['ok'] else 'OVER'
        extra = ('  raw moments %.2e' % r['raw_moments_err']) if 'raw_moments_err' in r else ''
        print('%s %-6s %-110s err=%.3e tol=%.3e%s' % (flag, tag, r['name'][:110], r['err'], r['tol'], extra), flush=True)
        table.setdefault(r['name'], {})[tag] = dict(err=r['err'], tol=r['tol'], ok=bool(r['ok']), raw=r.get('raw_moments_err'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--r152', action='store_true')
    ap.add_argument('--randbn', action='store_true')
    ap.add_argument('--out', default='gpurun_out/pivot_report.json')
    args = ap.parse_args()
    table = {}
    for case in [(64, 28, 256, 128, 1, 1), (64, 28, 128, 128, 3, 1), (512, 1, 2048, 128, 1, 1)]:
        for off in (30.0, 300.0, 3000.0):
            show('kernel', gc.check_conv_pivoted_stats(*case, offset=off), table)
    for name, on in (('pivot', '1'), ('raw', '0')):
        os.environ['SIMCLR_BN_PIVOT'] = on
        if args.r152:
            show(name, gc.check_train_step(depth=152, image_size=64, batch=4, compute_dtype='f32', num_classes=10, randomize_bn=False,
                                           sk_ratio=0.0625, width_multiplier=3, inputs='iid'), table)
            torch.cuda.empty_cache()
        if args.randbn:
            show(name, gc.check_train_step_fixed(depth=50, image_size=224, batch=32, compute_dtype='f32', randomize_bn=True), table)
            torch.cuda.empty_cache()
    os.environ.pop('SIMCLR_BN_PIVOT', None)
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(table, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
