"""A/B of the experimental multi-tap 3x3 wgrad kernel (SIMCLR_WGRAD_3X3=1) on the ResNet-50 stride-1 3x3 shapes:
time and max abs difference against the default kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simclr_amd import ops
from tools.microbench import timeit

def sweep_blocks(blocks):
    """nine-tap kernel only: time per layer for each total-workgroup target (SIMCLR_WGRAD3_BLOCKS)"""
    dev, dt, V = 'cuda', torch.bfloat16, 1024
    os.environ['SIMCLR_WGRAD_3X3'] = '1'
    print('%-14s ' % 'layer' + ' '.join('%7d' % b for b in blocks))
    for (H, C, cnt) in [(56, 64, 3), (28, 128, 3), (14, 256, 5), (7, 512, 2)]:
        x = torch.randn(V, H, H, C, device=dev).to(dt)
        dy = torch.randn(V, H, H, C, device=dev).to(dt)
        dw = torch.empty(9 * C, C, device=dev)
        ts = []
        for b in blocks:
            os.environ['SIMCLR_WGRAD3_BLOCKS'] = str(b)
            ts.append(timeit(lambda: ops.conv2d_wgrad(x, dy, 3, 3, 1, 1, out=dw), 5))
        print('%-14s ' % ('%dx%d C%d x%d' % (H, H, C, cnt)) + ' '.join('%7.0f' % t for t in ts), flush=True)
    os.environ.pop('SIMCLR_WGRAD3_BLOCKS')


def main():
    if '--blocks' in sys.argv:
        return sweep_blocks([int(b) for b in sys.argv[sys.argv.index('--blocks') + 1].split(',')])
    dev, dt, V = 'cuda', torch.bfloat16, 1024
    for (H, C, cnt) in [(56, 64, 3), (28, 128, 3), (14, 256, 5), (7, 512, 2)] + ([(56, 128, 0), (28, 256, 0), (14, 512, 0)] if '--wide' in sys.argv else []):
        x = torch.randn(V, H, H, C, device=dev).to(dt)
        dy = torch.randn(V, H, H, C, device=dev).to(dt)
        res = {}
        for flag in ('0', '1'):
            os.environ['SIMCLR_WGRAD_3X3'] = flag
            dw = torch.empty(9 * C, C, device=dev)
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, 3, 3, 1, 1, out=dw), 5)
            res[flag] = (t, dw.clone())
        os.environ['SIMCLR_WGRAD_3X3'] = '0'
        diff = float((res['1'][1] - res['0'][1]).abs().max()); scale = float(res['0'][1].abs().max())
        fl = 2.0 * V * H * H * 9 * C * C
        print('%dx%d C%d x%d: default %6.0f us (%4.0f TF/s)  multi-tap %6.0f us (%4.0f TF/s)  max|diff| %.3e of %.3e' % (
            H, H, C, cnt, res['0'][0], fl / res['0'][0] / 1e6, res['1'][0], fl / res['1'][0] / 1e6, diff, scale), flush=True)

if __name__ == '__main__':
    main()
