"""The ResNet-50 / 224 px / batch-32 fixed-gate step (tests/gpu_checks.py check_train_step_fixed) in every fp32 matrix
arithmetic mode, one process (the float64 oracle step is computed once):
    python tools/step_modes.py [--modes exact,bf16x3,bf16x6,bf16x6_3] [--batch 32] [--size 224] [--out gpurun_out/step_modes.json]
Prints one row per gate and mode; nothing is asserted (measurement tool for DESIGN.md section 5)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests import gpu_checks as gc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--modes', default='exact,bf16x3,bf16x6,bf16x6_3')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--inputs', default='structured')
    ap.add_argument('--pretrain_steps', type=int, default=0)
    ap.add_argument('--out', default='gpurun_out/step_modes.json')
    args = ap.parse_args()
    table = {}
    for mode in args.modes.split(','):
        t = time.time()
        kw = dict(depth=args.depth, image_size=args.size, batch=args.batch, inputs=args.inputs, pretrain_steps=args.pretrain_steps)
        if mode in ('bf16', 'bf16@unfused'):
            # 'bf16@unfused': SIMCLR_CONV3_FUSED=0 (every bottleneck tail as conv3 -> HBM -> bn_apply)
            if mode == 'bf16@unfused':
                os.environ['SIMCLR_CONV3_FUSED'] = '0'
            try:
                res = gc.check_train_step_fixed(compute_dtype='bf16', **kw)
            finally:
                os.environ.pop('SIMCLR_CONV3_FUSED', None)
        else:
            res = gc.check_train_step_fixed(compute_dtype='f32', f32_matmul=mode, **kw)
        for r in res:
            gate = r['name'].split(' ')[0]
            table.setdefault(gate, {})[mode] = (r['err'], r['tol'], r['ok'])
        print('mode %s: %.1f s' % (mode, time.time() - t), flush=True)
        torch.cuda.empty_cache()
    modes = args.modes.split(',')
    print('%-36s' % 'gate' + ''.join('%22s' % m for m in modes))
    for gate, row in table.items():
        print('%-36s' % gate + ''.join('%14.3e (%5.0e)%s' % (row[m][0], row[m][1], ' ' if row[m][2] else '!') if m in row else ' ' * 22 for m in modes))
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(table, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
