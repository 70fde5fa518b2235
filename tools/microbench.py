"""Per-layer micro-benchmark of the hot kernels at ResNet-50 1x / 224 px / V views per GPU.

python tools/microbench.py [--views 1024] [--dtype bf16] [--what conv,bn,ntxent,lars]
Prints one line per distinct layer shape: time (us), TFLOP/s, algorithmic GB/s; and a per-step
total weighted by how often the shape occurs.  Timing: HIP events on the launch stream, median of
`--iters` launches after warm-up; inputs are random (never zeros: DVFS).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from simclr_amd import ops  # noqa: E402

# (H, Cin, Cout, k, stride, count) of every conv in ResNet-50 1x after the stem (tf2/resnet.py:385-526)
R50 = [
    (56, 64, 256, 1, 1, 1),    # g1 shortcut
    (56, 64, 64, 1, 1, 1), (56, 256, 64, 1, 1, 2), (56, 64, 64, 3, 1, 3), (56, 64, 256, 1, 1, 3),
    (56, 256, 512, 1, 2, 1),   # g2 shortcut
    (56, 256, 128, 1, 1, 1), (56, 128, 128, 3, 2, 1), (28, 512, 128, 1, 1, 3), (28, 128, 128, 3, 1, 3),
    (28, 128, 512, 1, 1, 4),
    (28, 512, 1024, 1, 2, 1),  # g3 shortcut
    (28, 512, 256, 1, 1, 1), (28, 256, 256, 3, 2, 1), (14, 1024, 256, 1, 1, 5), (14, 256, 256, 3, 1, 5),
    (14, 256, 1024, 1, 1, 6),
    (14, 1024, 2048, 1, 2, 1),  # g4 shortcut
    (14, 1024, 512, 1, 1, 1), (14, 512, 512, 3, 2, 1), (7, 2048, 512, 1, 1, 2), (7, 512, 512, 3, 1, 2),
    (7, 512, 2048, 1, 1, 3),
]


def timeit(fn, iters, warm=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=1024)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--what', default='conv,bn,ntxent,lars')
    ap.add_argument('--out', default='gpurun_out/microbench.json')
    ap.add_argument('--ps', action='store_true', help='--dtype f32 with three backward terms: time dgrad / wgrad on a pre-split gradient operand')
    ap.add_argument('--f32_matmul', default='exact', help="--dtype f32: matrix arithmetic (ops.set_f32_matmul), e.g. bf16x6_3 = the parity mode")
    args = ap.parse_args()
    if args.dtype != 'bf16':
        ops.set_f32_matmul(args.f32_matmul)
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    dev = 'cuda'
    V = args.views
    what = args.what.split(',')
    res = []
    tot = dict(fwd=0.0, fwd_nostats=0.0, dgrad=0.0, wgrad=0.0, bn_apply=0.0, bn_bwd=0.0)
    if 'conv' in what:
        print('%-26s %9s %9s %9s %9s | TF/s fwd dgrad wgrad | GB/s fwd' % ('layer', 'fwd_us', 'nostat_us', 'dgrad_us', 'wgrad_us'))
        for (H, Cin, Cout, k, s, cnt) in R50:
            pad = (k - 1) // 2
            OH = (H + (k - 1) - k) // s + 1
            x = torch.randn(V, H, H, Cin, device=dev).to(dt)
            w = (torch.randn(k, k, Cin, Cout, device=dev) * (k * k * Cin) ** -0.5)
            dy = torch.randn(V, OH, OH, Cout, device=dev).to(dt)
            w_t = ops.prep_weights(w, 0, dt); w_d = ops.prep_weights(w, 1, dt)
            y = torch.empty(V, OH, OH, Cout, device=dev, dtype=dt)
            dx = torch.empty(V, H, H, Cin, device=dev, dtype=dt)
            dw = torch.empty(k * k * Cin, Cout, device=dev)
            stats = ops.conv_stats(V * OH * OH, Cout, dev)
            if args.ps and dt == torch.float32 and Cout % 32 == 0 and Cin % 64 == 0:
                # the gradient operand as the BatchNorm backward hands it over in the parity mode: pre-split (hi, lo) bf16 pieces
                one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
                dy, _ = ops.bn_bwd_apply(dy, dy, None, one, zero, zero, one, zero, zero, 0, ps_out=True)
            t_f = timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=stats, out=y), args.iters)
            t_n = timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=None, out=y), args.iters)
            t_d = timeit(lambda: ops.conv2d_dgrad(dy, w_d, k, k, s, pad, H, H, out=dx), args.iters)
            t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, k, k, s, pad, out=dw), args.iters)
            t_db = 0.0
            if s == 1:      # dgrad with the fused BatchNorm-backward reduce (mask recomputed from the BN input: mode 2)
                bn = dict(x=x, mask=None, scale=torch.rand(Cin, device=dev) - 0.4, shift=torch.randn(Cin, device=dev) * 0.3,
                          mean=torch.randn(Cin, device=dev) * 0.2, rstd=torch.rand(Cin, device=dev) + 0.5, mode=2)
                t_db = timeit(lambda: ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, H, bn, out=dx), args.iters)
            fl = 2.0 * V * OH * OH * k * k * Cin * Cout
            by = x.element_size() * (x.numel() + y.numel())
            name = '%dx%d %d->%d k%d s%d x%d' % (H, H, Cin, Cout, k, s, cnt)
            print('%-26s %9.0f %9.0f %9.0f %9.0f | %6.0f %6.0f %6.0f | %6.0f | dgrad_bn %4.0f' % (
                name, t_f, t_n, t_d, t_w, fl / t_f / 1e6, fl / t_d / 1e6, fl / t_w / 1e6, by / t_f / 1e3, t_db), flush=True)
            res.append(dict(layer=name, fwd_us=t_f, fwd_nostats_us=t_n, dgrad_us=t_d, wgrad_us=t_w, dgrad_bn_us=t_db, flops=fl,
                            bytes=by, count=cnt))
            tot['fwd'] += cnt * t_f; tot['fwd_nostats'] += cnt * t_n; tot['dgrad'] += cnt * t_d; tot['wgrad'] += cnt * t_w
            del x, dy, y, dx
        print('per-step totals (ms): fwd %.2f (no stats %.2f)  dgrad %.2f  wgrad %.2f' % (
            tot['fwd'] / 1e3, tot['fwd_nostats'] / 1e3, tot['dgrad'] / 1e3, tot['wgrad'] / 1e3), flush=True)
    if 'tile' in what:
        # A/B of the forward / dgrad tile: default (128-wide) vs SIMCLR_IGEMM_TILE=256 on every layer whose width allows it
        print('%-26s | fwd+stats 128 / 256 | fwd 128 / 256 | dgrad 128 / 256 | dgrad_bn 128 / 256  (us)' % 'layer')
        tot = [0.0] * 8
        for (H, Cin, Cout, k, s, cnt) in R50:
            if Cout % 256 and Cin % 256:
                continue
            pad = (k - 1) // 2
            OH = (H + (k - 1) - k) // s + 1
            x = torch.randn(V, H, H, Cin, device=dev).to(dt)
            w = (torch.randn(k, k, Cin, Cout, device=dev) * (k * k * Cin) ** -0.5)
            dy = torch.randn(V, OH, OH, Cout, device=dev).to(dt)
            w_t = ops.prep_weights(w, 0, dt); w_d = ops.prep_weights(w, 1, dt)
            y = torch.empty(V, OH, OH, Cout, device=dev, dtype=dt)
            dx = torch.empty(V, H, H, Cin, device=dev, dtype=dt)
            stats = ops.conv_stats(V * OH * OH, Cout, dev)
            bn = dict(x=x, mask=None, scale=torch.rand(Cin, device=dev) - 0.4, shift=torch.randn(Cin, device=dev) * 0.3,
                      mean=torch.randn(Cin, device=dev) * 0.2, rstd=torch.rand(Cin, device=dev) + 0.5, mode=2)
            ts = []
            for tile in ('128', '256'):
                os.environ['SIMCLR_IGEMM_TILE'] = tile
                ts.append(timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=stats, out=y), args.iters))
                ts.append(timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=None, out=y), args.iters))
                ts.append(timeit(lambda: ops.conv2d_dgrad(dy, w_d, k, k, s, pad, H, H, out=dx), args.iters))
                ts.append(timeit(lambda: ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, H, bn, out=dx), args.iters) if s == 1 else 0.0)
            os.environ.pop('SIMCLR_IGEMM_TILE')
            name = '%dx%d %d->%d k%d s%d x%d' % (H, H, Cin, Cout, k, s, cnt)
            print('%-26s | %6.0f %6.0f | %6.0f %6.0f | %6.0f %6.0f | %6.0f %6.0f' % (
                name, ts[0], ts[4], ts[1], ts[5], ts[2], ts[6], ts[3], ts[7]), flush=True)
            res.append(dict(layer='tile ' + name, t128=ts[:4], t256=ts[4:], count=cnt))
            for i in range(8):
                tot[i] += cnt * ts[i]
            del x, dy, y, dx
        print('weighted totals (ms): fwd+stats %.2f / %.2f  fwd %.2f / %.2f  dgrad %.2f / %.2f  dgrad_bn %.2f / %.2f' % (
            tot[0] / 1e3, tot[4] / 1e3, tot[1] / 1e3, tot[5] / 1e3, tot[2] / 1e3, tot[6] / 1e3, tot[3] / 1e3, tot[7] / 1e3), flush=True)
    if 'wide' in what:
        # A/B of the forward / dgrad schedules, us per launch: 128-wide persistent tiles without / with the split tail, the
        # eight-phase 256 x 256 tile (csrc/igemm_wide.h) without / with it.  Columns: fwd+stats, dgrad, dgrad + BN reduce.
        cfgs = [('narrow', dict(SIMCLR_IGEMM_SPLIT='0')), ('narrow+split', {}), ('wide', dict(SIMCLR_IGEMM_WIDE='2', SIMCLR_IGEMM_SPLIT='0')),
                ('wide+split', dict(SIMCLR_IGEMM_WIDE='2'))]
        print('%-26s | %s' % ('layer', ' | '.join('%-22s' % c[0] for c in cfgs)))
        tot = {c[0]: [0.0, 0.0, 0.0] for c in cfgs}
        for (H, Cin, Cout, k, s, cnt) in R50:
            if Cout % 256 and Cin % 256:
                continue
            pad = (k - 1) // 2
            OH = (H + (k - 1) - k) // s + 1
            x = torch.randn(V, H, H, Cin, device=dev).to(dt)
            w = (torch.randn(k, k, Cin, Cout, device=dev) * (k * k * Cin) ** -0.5)
            dy = torch.randn(V, OH, OH, Cout, device=dev).to(dt)
            w_t = ops.prep_weights(w, 0, dt); w_d = ops.prep_weights(w, 1, dt)
            y = torch.empty(V, OH, OH, Cout, device=dev, dtype=dt)
            dx = torch.empty(V, H, H, Cin, device=dev, dtype=dt)
            stats = ops.conv_stats(V * OH * OH, Cout, dev)
            bn = dict(x=x, mask=None, scale=torch.rand(Cin, device=dev) - 0.4, shift=torch.randn(Cin, device=dev) * 0.3,
                      mean=torch.randn(Cin, device=dev) * 0.2, rstd=torch.rand(Cin, device=dev) + 0.5, mode=2)
            row = {}
            for name, env in cfgs:
                for kk, vv in env.items():
                    os.environ[kk] = vv
                ts = [timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=stats, out=y), args.iters),
                      timeit(lambda: ops.conv2d_dgrad(dy, w_d, k, k, s, pad, H, H, out=dx), args.iters),
                      timeit(lambda: ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, H, bn, out=dx), args.iters) if s == 1 else 0.0]
                for kk in env:
                    os.environ.pop(kk)
                row[name] = ts
                for i in range(3):
                    tot[name][i] += cnt * ts[i]
            name = '%dx%d %d->%d k%d s%d x%d' % (H, H, Cin, Cout, k, s, cnt)
            print('%-26s | %s' % (name, ' | '.join('%6.0f %6.0f %6.0f  ' % tuple(row[c[0]]) for c in cfgs)), flush=True)
            res.append(dict(layer='wide ' + name, count=cnt, **row))
            del x, dy, y, dx
        print('%-26s | %s' % ('weighted totals (ms)', ' | '.join('%6.2f %6.2f %6.2f  ' % tuple(v / 1e3 for v in tot[c[0]]) for c in cfgs)), flush=True)
    if 'comparator' in what:
        # the vendor libraries on the SAME shapes, as a yardstick only (never on the product path):
        # MIOpen through torch.nn.functional.conv2d (bf16, channels_last = NHWC memory) forward / data gradient / weight
        # gradient, and hipBLASLt through torch.matmul for the 1x1 stride-1 layers (the same GEMMs without the conv wrapper).
        import torch.nn.functional as F
        torch.backends.cudnn.benchmark = os.environ.get('SIMCLR_CMP_FIND', '0') == '1'     # default: MIOpen immediate mode (heuristic pick)
        print('%-26s | ours fwd dgrad wgrad | MIOpen fwd dgrad wgrad | hipBLASLt fwd dgrad wgrad | bound us (MFMA / HBM@8)' % 'layer')
        for (H, Cin, Cout, k, s, cnt) in R50:
            pad = (k - 1) // 2
            OH = (H + (k - 1) - k) // s + 1
            x = torch.randn(V, H, H, Cin, device=dev).to(dt)
            w = (torch.randn(k, k, Cin, Cout, device=dev) * (k * k * Cin) ** -0.5)
            dy = torch.randn(V, OH, OH, Cout, device=dev).to(dt)
            w_t = ops.prep_weights(w, 0, dt); w_d = ops.prep_weights(w, 1, dt)
            y = torch.empty(V, OH, OH, Cout, device=dev, dtype=dt)
            dx = torch.empty(V, H, H, Cin, device=dev, dtype=dt)
            dw = torch.empty(k * k * Cin, Cout, device=dev)
            ours = [timeit(lambda: ops.conv2d_fwd(x, w_t, k, k, s, pad, OH, OH, stats=None, out=y), args.iters),
                    timeit(lambda: ops.conv2d_dgrad(dy, w_d, k, k, s, pad, H, H, out=dx), args.iters),
                    timeit(lambda: ops.conv2d_wgrad(x, dy, k, k, s, pad, out=dw), args.iters)]
            xn = x.permute(0, 3, 1, 2)                       # NCHW view of the NHWC tensor = channels_last
            dyn = dy.permute(0, 3, 1, 2)
            wn = w.permute(3, 2, 0, 1).to(dt).contiguous(memory_format=torch.channels_last)
            mi = [float('nan')] * 3
            try:
                mi[0] = timeit(lambda: F.conv2d(xn, wn, None, s, pad), args.iters)
                mi[1] = timeit(lambda: torch.ops.aten.convolution_backward(dyn, xn, wn, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]), args.iters)
                mi[2] = timeit(lambda: torch.ops.aten.convolution_backward(dyn, xn, wn, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]), args.iters)
            except Exception as e:  # noqa: BLE001
                print('   MIOpen failed on this shape:', repr(e)[:200])
            bl = [float('nan')] * 3
            if k == 1 and s == 1:
                A = x.view(-1, Cin); G = dy.view(-1, Cout); W2 = w.view(Cin, Cout).to(dt); W2t = W2.t().contiguous()
                bl = [timeit(lambda: torch.matmul(A, W2), args.iters), timeit(lambda: torch.matmul(G, W2t), args.iters),
                      timeit(lambda: torch.matmul(A.t(), G), args.iters)]
            fl = 2.0 * V * OH * OH * k * k * Cin * Cout
            by = x.element_size() * (x.numel() + y.numel() + w.numel())
            name = '%dx%d %d->%d k%d s%d x%d' % (H, H, Cin, Cout, k, s, cnt)
            print('%-26s | %5.0f %5.0f %5.0f | %5.0f %5.0f %5.0f | %5.0f %5.0f %5.0f | %4.0f / %4.0f' % (
                name, *ours, *mi, *bl, fl / 2.5e9, by / 8e6), flush=True)
            res.append(dict(layer='cmp ' + name, ours=ours, miopen=mi, hipblaslt=bl, flops=fl, bytes=by, count=cnt))
            del x, dy, y, dx, xn, dyn
    if 'bn' in what:
        print('%-22s %9s %9s %9s %9s | GB/s apply resid bwd_red bwd_app' % ('tensor', 'apply_us', 'resid_us', 'bwdred_us', 'bwdapp_us'))
        shapes = [(56, 64, 7), (56, 256, 4), (28, 128, 8), (28, 512, 5), (14, 256, 12), (14, 1024, 7), (7, 512, 6), (7, 2048, 4)]
        for (H, C, cnt) in shapes:
            x = torch.randn(V, H, H, C, device=dev).to(dt)
            r = torch.randn(V, H, H, C, device=dev).to(dt)
            dy = torch.randn(V, H, H, C, device=dev).to(dt)
            y = torch.empty_like(x); dx = torch.empty_like(x)
            scale = torch.rand(C, device=dev) + 0.5; shift = torch.randn(C, device=dev) * 0.1
            mean = torch.randn(C, device=dev) * 0.1; rstd = torch.rand(C, device=dev) + 0.5
            c1 = torch.randn(C, device=dev) * 0.01; c2 = torch.randn(C, device=dev) * 0.01
            t_a = timeit(lambda: ops.bn_apply(x, scale, shift, True, out=y), args.iters)
            t_r = timeit(lambda: ops.bn_apply(x, scale, shift, True, res=r, out=y), args.iters)
            t_br = timeit(lambda: ops.bn_bwd_reduce(dy, x, y, scale, shift, mean, rstd, 1), args.iters)
            t_ba = timeit(lambda: ops.bn_bwd_apply(dy, x, y, scale, shift, mean, rstd, c1, c2, 1, out=dx), args.iters)
            nb = x.numel() * x.element_size()
            print('%-22s %9.0f %9.0f %9.0f %9.0f | %6.0f %6.0f %6.0f %6.0f' % (
                '%dx%d C%d x%d' % (H, H, C, cnt), t_a, t_r, t_br, t_ba, 2 * nb / t_a / 1e3, 3 * nb / t_r / 1e3,
                3 * nb / t_br / 1e3, 4 * nb / t_ba / 1e3), flush=True)
            res.append(dict(layer='bn %dx%d C%d' % (H, H, C), apply_us=t_a, resid_us=t_r, bwdred_us=t_br, bwdapp_us=t_ba, bytes=nb, count=cnt))
            del x, r, dy, y, dx
    if 'ntxent' in what:
        for (n, N) in [(512, 512), (512, 4096), (256, 2048), (4096, 4096)]:
            D = 128
            zl = torch.nn.functional.normalize(torch.randn(2 * n, D, device=dev), dim=1)
            za = torch.nn.functional.normalize(torch.randn(2 * N, D, device=dev), dim=1)
            za[:n] = zl[:n]; za[N:N + n] = zl[n:]
            ws = ops.ntxent_workspace(n, N, D, dev)
            fl = 24.0 * n * N * D
            by = 2.0 * (2 * n + 2 * N) * D * 4
            for split in (False, True):        # exact fp32-input MFMA sweeps | the opt-in split-fp16 sweeps (FLAGS.ntxent_matmul='f16x3')
                out, rs, _ = ops.ntxent_fwd(zl, za, 0, 0.1, ws, split=split)
                t_f = timeit(lambda: ops.ntxent_fwd(zl, za, 0, 0.1, ws, split=split), args.iters)
                t_b = timeit(lambda: ops.ntxent_bwd(zl, za, 0, 0.1, rs, 1.0, out, ws, split=split), args.iters)
                print('ntxent n=%d N=%d %s: fwd %.0f us bwd %.0f us | fused fwd+bwd %.1f TF/s (24nND), algorithmic %.1f GB/s' % (
                    n, N, 'f16x3' if split else 'exact', t_f, t_b, fl / (t_f + t_b) / 1e6, by / (t_f + t_b) / 1e3), flush=True)
                res.append(dict(layer='ntxent n%d N%d%s' % (n, N, ' f16x3' if split else ''), fwd_us=t_f, bwd_us=t_b, flops=fl, bytes=by))
    if 'lars' in what:
        from simclr_amd.lars_optimizer import LARSOptimizer, Variable
        sizes = []
        for (H, Cin, Cout, k, s, cnt) in R50:
            sizes += [(k * k * Cin * Cout,)] * cnt + [(Cout,), (Cout,)] * cnt
        sizes += [(7 * 7 * 3 * 64,), (2048 * 2048,), (2048 * 2048,), (2048 * 128,), (2048 * 1000,), (1000,)]
        vs = []
        for i, shp in enumerate(sizes):
            v = Variable('conv2d_%d/kernel:0' % i if shp[0] > 4096 else 'batch_normalization_%d/gamma:0' % i,
                         torch.randn(shp, device=dev) * 0.05)
            v.grad = torch.randn(shp, device=dev) * 1e-3
            vs.append(v)
        opt = LARSOptimizer(0.1, weight_decay=1e-6, exclude_from_weight_decay=['batch_normalization', 'bias', 'head_supervised'])
        gv = [(v.grad, v) for v in vs]
        t = timeit(lambda: opt.apply_gradients(gv), args.iters)
        nel = sum(s[0] for s in sizes)
        print('lars %d tensors %.1f M elems: %.0f us, %.0f GB/s (7 x 4 B/elem two-pass)' % (len(vs), nel / 1e6, t, 28.0 * nel / t / 1e3), flush=True)
        res.append(dict(layer='lars', us=t, elems=nel))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
