"""Time attribution for the conv kernels (diagnostic library only).

  bash simclr_amd/csrc/build.sh diag
  SIMCLR_HIP_LIB=simclr_amd/libsimclr_hip_diag.so python tools/diag_conv.py

For a few representative ResNet-50 layers, times fwd(+BN stats) / dgrad(+fused BN-backward epilogue) / wgrad with
parts of the kernel switched off (env SIMCLR_DIAG, read per launch by the diagnostic build):
  igemm:  1 = no LDS reads + MFMA   2 = no global->LDS loads   4 = no epilogue at all
          8 = epilogue operand loads all hit one cache line    16 = no epilogue stores
  wgrad:  1 = no LDS reads + MFMA   2 = no global loads        4 = no LDS stores
Outputs of the switched-off runs are wrong by design; only the times mean anything.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from simclr_amd import ops  # noqa: E402
from simclr_amd import _lib  # noqa: E402

LAYERS = [(56, 64, 256, 1), (56, 256, 64, 1), (56, 64, 64, 3), (28, 128, 512, 1), (28, 128, 128, 3),
          (14, 256, 1024, 1), (14, 1024, 256, 1), (14, 256, 256, 3), (7, 512, 512, 3)]
IGEMM_MODES = [0, 1, 2, 3, 4, 8, 16, 7]
WGRAD_MODES = [0, 1, 2, 4, 6, 3]


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def sweep(fn, modes):
    out = []
    for m in modes:
        os.environ['SIMCLR_DIAG'] = str(m)
        out.append(timeit(fn))
    os.environ['SIMCLR_DIAG'] = '0'
    return out


def wgrad_cfg_sweep():
    """Every distinct ResNet-50 conv shape: wgrad time under each kernel variant (SIMCLR_WGRAD_CFG 0..3)."""
    from tools.microbench import R50
    dev, dt, V = 'cuda', torch.bfloat16, 1024
    variants = [('128 auto', 1, 0, -1), ('128 xcd0', 1, 0, 0), ('128 xcd1', 1, 0, 1), ('256 xcd0', 1, 1, 0), ('256 xcd1', 1, 1, 1)]
    tot = [0.0] * len(variants)
    print('%-26s ' % 'wgrad layer' + ' '.join('%9s' % v[0] for v in variants))
    for (H, Cin, Cout, k, s, cnt) in R50:
        pad = (k - 1) // 2
        OH = (H + (k - 1) - k) // s + 1
        x = torch.randn(V, H, H, Cin, device=dev).to(dt)
        dy = torch.randn(V, OH, OH, Cout, device=dev).to(dt)
        dw = torch.empty(k * k * Cin, Cout, device=dev)
        ts = []
        for vi, (_, cfg, t256, xcd) in enumerate(variants):
            os.environ['SIMCLR_WGRAD_CFG'] = str(cfg)
            os.environ['SIMCLR_WGRAD_256'] = str(t256)
            os.environ['SIMCLR_WGRAD_XCD'] = str(xcd)
            ts.append(timeit(lambda: ops.conv2d_wgrad(x, dy, k, k, s, pad, out=dw)))
            tot[vi] += cnt * ts[-1]
        print('%-26s ' % ('%dx%d %d->%d k%d s%d x%d' % (H, H, Cin, Cout, k, s, cnt)) + ' '.join('%9.0f' % t for t in ts), flush=True)
        del x, dy
    os.environ.pop('SIMCLR_WGRAD_CFG'); os.environ.pop('SIMCLR_WGRAD_256'); os.environ.pop('SIMCLR_WGRAD_XCD')
    print('per-step totals (ms): ' + ' '.join('%.2f' % (t / 1e3) for t in tot), flush=True)


def main():
    assert 'diag' in _lib.LIB_PATH
    if '--wgrad' in sys.argv:
        return wgrad_cfg_sweep(), 'run with SIMCLR_HIP_LIB=simclr_amd/libsimclr_hip_diag.so'
    dev, dt, V = 'cuda', torch.bfloat16, 1024
    print('igemm modes', IGEMM_MODES, ' wgrad modes', WGRAD_MODES)
    for (H, Cin, Cout, k) in LAYERS:
        pad = (k - 1) // 2
        x = torch.randn(V, H, H, Cin, device=dev).to(dt)
        xo = torch.randn(V, H, H, Cin, device=dev).to(dt)
        w = torch.randn(k, k, Cin, Cout, device=dev) * (k * k * Cin) ** -0.5
        dy = torch.randn(V, H, H, Cout, device=dev).to(dt)
        w_t = ops.prep_weights(w, 0, dt); w_d = ops.prep_weights(w, 1, dt)
        y = torch.empty(V, H, H, Cout, device=dev, dtype=dt)
        dx = torch.randn(V, H, H, Cin, device=dev).to(dt)
        dw = torch.empty(k * k * Cin, Cout, device=dev)
        stats = ops.conv_stats(V * H * H, Cout, dev)
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
        mean = torch.randn(Cin, device=dev) * 0.1; rstd = torch.rand(Cin, device=dev) + 0.5
        bn2 = dict(x=x, mask=None, scale=sc, shift=sh, mean=mean, rstd=rstd, mode=2)
        bn1 = dict(x=x, mask=xo, scale=None, shift=None, mean=mean, rstd=rstd, mode=1)
        name = '%dx%d %d->%d k%d' % (H, H, Cin, Cout, k)
        fl = 2.0 * V * H * H * k * k * Cin * Cout
        rows = [
            ('fwd+stats', sweep(lambda: ops.conv2d_fwd(x, w_t, k, k, 1, pad, H, H, stats=stats, out=y), IGEMM_MODES)),
            ('dgrad', sweep(lambda: ops.conv2d_dgrad(dy, w_d, k, k, 1, pad, H, H, out=dx), IGEMM_MODES)),
            ('dgrad+acc', sweep(lambda: ops.conv2d_dgrad(dy, w_d, k, k, 1, pad, H, H, out=dx, accumulate=True), IGEMM_MODES)),
            ('dgrad_bn m2', sweep(lambda: ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, H, bn2, out=dx), IGEMM_MODES)),
            ('dgrad_bn m1+acc', sweep(lambda: ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, H, bn1, out=dx, accumulate=True), IGEMM_MODES)),
            ('wgrad', sweep(lambda: ops.conv2d_wgrad(x, dy, k, k, 1, pad, out=dw), WGRAD_MODES)),
        ]
        for (op, ts) in rows:
            print('%-20s %-16s %s   | %5.0f TF/s' % (name, op, ' '.join('%6.0f' % t for t in ts), fl / ts[0] / 1e6), flush=True)
        del x, xo, dy, y, dx


if __name__ == '__main__':
    main()
