"""Kernel-level parity checks: HIP path (through the C ABI) vs CPU references.

Each check returns a dict {name, err, tol, ok, ...}.  The CPU references are the
oracle (`oracle/`, float64) for the loss / LARS and plain torch-CPU float64 ops
for conv / BN / pooling.  Used by tests/test_gpu_kernels.py (pytest -m gpu) and
tools/run_gpu_checks.py (prints everything; first-contact diagnostics).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import lars as olars
from oracle import ntxent as ont
from simclr_amd import ops
from simclr_amd._lib import lib

DEV = 'cuda'


def _res(name, got, ref, rtol, atol=0.0, **kw):
    got = torch.as_tensor(got).detach().double().cpu()
    ref = torch.as_tensor(ref).detach().double().cpu()
    err = float((got - ref).abs().max()) if got.numel() else 0.0
    scale = float(ref.abs().max()) if ref.numel() else 0.0
    tol = rtol * scale + atol
    bad = int(((got - ref).abs() > tol).sum())
    d = dict(name=name, err=err, tol=tol, scale=scale, ok=bool(err <= tol) and bool(torch.isfinite(got).all()),
             nbad=bad, numel=got.numel())
    d.update(kw)
    return d


def _tol(dtype):
    # bf16: inputs are pre-rounded to bf16, so the only error is fp32 accumulation order
    # plus one final bf16 rounding of the output (2^-9 relative) ; f32: accumulation order.
    return 6e-3 if dtype == torch.bfloat16 else 2e-5


# ------------------------------------------------------------------ probes
def check_probes():
    out = []
    g = torch.Generator().manual_seed(0)
    a = torch.randn(16, 32, generator=g).bfloat16()
    b = torch.randn(32, 16, generator=g).bfloat16()   # asymmetric on purpose
    o = torch.zeros(64 * 4, device=DEV)
    ad, bd = a.to(DEV).view(torch.int16), b.to(DEV).view(torch.int16)   # keep alive across the launch
    lib().probe(0, ops._p(ad), ops._p(bd), ops._p(o), ops._s())
    d = o.cpu().view(64, 4)
    ref = a.double() @ b.double()
    got = torch.zeros(16, 16, dtype=torch.float64)
    for l in range(64):
        for r in range(4):
            got[(l >> 4) * 4 + r, l & 15] = d[l, r]
    out.append(_res('probe_mfma_bf16_16x16x32_layout', got, ref, 1e-5))
    a = torch.randn(16, 4, generator=g)
    b = torch.randn(4, 16, generator=g)
    o = torch.zeros(64 * 4, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    lib().probe(1, ops._p(ad), ops._p(bd), ops._p(o), ops._s())
    d = o.cpu().view(64, 4)
    got = torch.zeros(16, 16, dtype=torch.float64)
    for l in range(64):
        for r in range(4):
            got[(l >> 4) * 4 + r, l & 15] = d[l, r]
    out.append(_res('probe_mfma_f32_16x16x4_layout', got, a.double() @ b.double(), 1e-6))
    # v_mfma_f32_16x16x32_f16: the bf16 layout, and SUBNORMAL fp16 inputs must take part in the products (the lo pieces of the
    # split-fp16 forward, csrc/conv.hip split_pair<true>, fall below 2^-14 for |x| < 2^-3)
    for tag, sa in (('normal', 1.0), ('subnormal_a', 2.0 ** -18)):
        a = (torch.randn(16, 32, generator=g) * sa).half()
        b = torch.randn(32, 16, generator=g).half()
        o = torch.zeros(64 * 4, device=DEV)
        ad, bd = a.to(DEV).view(torch.int16), b.to(DEV).view(torch.int16)
        lib().probe(3, ops._p(ad), ops._p(bd), ops._p(o), ops._s())
        d = o.cpu().view(64, 4)
        ref = a.double() @ b.double()
        got = torch.zeros(16, 16, dtype=torch.float64)
        for l in range(64):
            for r in range(4):
                got[(l >> 4) * 4 + r, l & 15] = d[l, r]
        out.append(_res('probe_mfma_f16_16x16x32_%s' % tag, got, ref, 1e-5))
    return out


def probe_ds_read_tr16():
    """Returns the raw lane mapping of ds_read_b64_tr_b16 for a [*,16]-element row-major tile:
    lane l supplies address (l>>4)*64 + ((l&15)>>2)*16 + (l&3)*4 (elements)."""
    src = torch.arange(1024, dtype=torch.int16)
    lanes = torch.arange(64)
    addr = ((lanes >> 4) * 64 + ((lanes & 15) >> 2) * 16 + (lanes & 3) * 4).int()
    o = torch.zeros(256, dtype=torch.int16, device=DEV)
    sd, ad = src.to(DEV), addr.to(DEV)
    lib().probe(2, ops._p(sd), ops._p(ad), ops._p(o), ops._s())
    torch.cuda.synchronize()
    return o.cpu().view(64, 4)


# ------------------------------------------------------------------ NT-Xent
def check_ntxent(n, R, D=128, temperature=0.1, rank=0, seed=3, hidden_norm=True, split=False):
    """split: the sweeps' fp32 products as three fp16-piece MFMA terms (opt-in mode, l2-normalised rows) -- same gates."""
    g = np.random.default_rng(seed)
    hs = [g.standard_normal((2 * n, D)).astype(np.float32) for _ in range(R)]
    losses, grads = ont.contrastive_loss_and_grad(hs, hidden_norm, temperature)
    loss_r, logits_ab, labels = ont.add_contrastive_loss(hs[rank], hidden_norm, temperature,
                                                        all_hiddens=hs if R > 1 else None, replica_id=rank)
    acc_ref, ent_ref = ont.contrastive_metrics(logits_ab, labels)
    # device path: normalise every replica's hidden, build z_all, run rank's fwd/bwd
    zs, invs = [], []
    for h in hs:
        x = torch.from_numpy(h).to(DEV)
        if hidden_norm:
            z, inv = ops.l2norm_fwd(x)
        else:
            z, inv = x, None
        zs.append(z); invs.append(inv)
    z_all = torch.cat([z[:n] for z in zs] + [z[n:] for z in zs], 0).contiguous()
    res = []
    if hidden_norm:
        zref = ont.l2_normalize(hs[rank].astype(np.float64))
        res.append(_res('l2norm_fwd n=%d' % n, zs[rank], zref, 0, 1e-6))
    out, row_stats, ws = ops.ntxent_fwd(zs[rank], z_all, rank, temperature, split=split)
    # total gradient wrt rank's hidden = local part + sum over replicas q of dz_all_q[rank slot]
    dz_local_r = None
    dz_slot = torch.zeros(2 * n, D, device=DEV)
    for q in range(R):
        o_q, rs_q, ws_q = ops.ntxent_fwd(zs[q], z_all, q, temperature, split=split)
        dl, da = ops.ntxent_bwd(zs[q], z_all, q, temperature, rs_q, 1.0 / R, o_q, ws_q, split=split)
        if q == rank:
            dz_local_r = dl
            out = o_q
        N = n * R
        dz_slot[:n] += da[rank * n:(rank + 1) * n]
        dz_slot[n:] += da[N + rank * n:N + (rank + 1) * n]
    dz = dz_local_r + dz_slot
    dh = ops.l2norm_bwd(zs[rank], invs[rank], dz) if hidden_norm else dz
    torch.cuda.synchronize()
    o = out.cpu().double()
    tag = 'n=%d R=%d D=%d T=%g rank=%d%s' % (n, R, D, temperature, rank, ' f16x3' if split else '')
    res.append(_res('ntxent_loss ' + tag, o[0], loss_r, 1e-5))
    res.append(_res('ntxent_acc ' + tag, o[1], acc_ref, 0, 1e-6))
    res.append(_res('ntxent_entropy ' + tag, o[2], ent_ref, 1e-4, 1e-6))
    res.append(_res('ntxent_grad ' + tag, dh, grads[rank], 2e-4))
    lab = ops.ntxent_logits_ab(zs[rank], z_all, temperature)
    res.append(_res('ntxent_logits_ab ' + tag, lab, logits_ab, 1e-5))
    return res


def check_ntxent_closed_forms():
    """SURVEY section 4 known answers: (i) identical rows -> 2*log(2N-1); (ii) orthogonal one-hot."""
    res = []
    n, D, T = 64, 128, 0.1
    h = torch.ones(2 * n, D, device=DEV)
    z, _ = ops.l2norm_fwd(h)
    out, _, _ = ops.ntxent_fwd(z, z, 0, T)
    res.append(_res('ntxent_closed_identical', out.cpu()[0], 2 * np.log(2 * n - 1), 1e-5))
    e = torch.eye(n, D, device=DEV)
    h = torch.cat([e, e], 0).contiguous()
    z, _ = ops.l2norm_fwd(h)
    out, _, _ = ops.ntxent_fwd(z, z, 0, T)
    closed = 2 * (np.log(np.exp(1 / T) + (2 * n - 2)) - 1 / T)
    res.append(_res('ntxent_closed_orthogonal', out.cpu()[0], closed, 1e-4, 1e-7))
    return res


# ------------------------------------------------------------------ LARS
def check_lars(seed=0, classic=True, nesterov=False):
    from simclr_amd.lars_optimizer import LARSOptimizer, Variable
    g = np.random.default_rng(seed)
    shapes = [('conv2d/kernel:0', (3, 3, 64, 64)), ('batch_normalization/gamma:0', (64,)),
              ('dense/kernel:0', (300, 77)), ('head_supervised/linear_layer/dense/bias:0', (10,)),
              ('zero/kernel:0', (1000,)), ('conv2d_1/kernel:0', (1, 1, 256, 1030))]
    vs = []
    ref = {}
    for name, shp in shapes:
        w = (g.standard_normal(shp) * 0.05).astype(np.float32)
        if name.startswith('zero'):
            w[:] = 0
        gr = (g.standard_normal(shp) * 1e-3).astype(np.float32)
        m = (g.standard_normal(shp) * 1e-3).astype(np.float32)
        v = Variable(name, torch.from_numpy(w).to(DEV))
        v.grad = torch.from_numpy(gr).to(DEV)
        vs.append(v)
        ref[name] = (w, gr, m)
    opt = LARSOptimizer(0.3, momentum=0.9, weight_decay=1e-4, use_nesterov=nesterov, classic_momentum=classic,
                        exclude_from_weight_decay=['batch_normalization', 'bias', 'head_supervised'])
    opt._build(vs)
    for v in vs:
        opt.get_slot(v, 'Momentum').copy_(torch.from_numpy(ref[v.name][2]).to(DEV))
    opt.apply_gradients([(v.grad, v) for v in vs])
    torch.cuda.synchronize()
    res = []
    for v in vs:
        w, gr, m = ref[v.name]
        nw, nv = olars.lars_apply(v.name, w, gr, m, 0.3, momentum=0.9, weight_decay=1e-4, use_nesterov=nesterov,
                                  classic_momentum=classic,
                                  exclude_from_weight_decay=['batch_normalization', 'bias', 'head_supervised'])
        tag = 'lars[%s classic=%d nest=%d]' % (v.name, classic, nesterov)
        res.append(_res(tag + ' w', v.value, nw, 2e-6, 1e-9))
        res.append(_res(tag + ' v', opt.get_slot(v, 'Momentum'), nv, 2e-6, 1e-9))
    return res


# ------------------------------------------------------------------ conv
def _rand(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def check_conv(V, H, W, Cin, Cout, k, stride, dtype, seed=0, matmul='exact'):
    """matmul (fp32 only): 'exact' | 'bf16x3' | 'bf16x6' -- the split-bf16 arithmetic of simclr_set_f32_matmul.  Gates: a
    three-term product carries <= 3 * 2^-18 relative error (two representation residuals + the dropped lo*lo term), so
    against max|ref| the bound is 4e-5; six terms are held to the exact mode's 2e-5."""
    ops.set_f32_matmul(matmul)
    try:
        return _check_conv(V, H, W, Cin, Cout, k, stride, dtype, seed, matmul)
    finally:
        ops.set_f32_matmul('exact')


def _check_conv(V, H, W, Cin, Cout, k, stride, dtype, seed, matmul):
    g = torch.Generator().manual_seed(seed)
    pad = (k - 1) // 2
    OH = (H + (k - 1) - k) // stride + 1
    OW = (W + (k - 1) - k) // stride + 1
    x = _rand((V, H, W, Cin), dtype, g)
    w = _rand((k, k, Cin, Cout), dtype, g, (k * k * Cin) ** -0.5)     # HWIO, values representable in T
    dy = _rand((V, OH, OW, Cout), dtype, g)
    # CPU float64 reference with TF Conv2dFixedPadding semantics
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    pe = (k - 1) - pad
    yr = F.conv2d(F.pad(xr, (pad, pe, pad, pe)), wr.permute(3, 2, 0, 1), stride=stride)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    y_ref = yr.detach().permute(0, 2, 3, 1)
    dx_ref = xr.grad.permute(0, 2, 3, 1)
    dw_ref = wr.grad
    # device
    xd, dyd = x.to(DEV), dy.to(DEV)
    wd32 = w.float().to(DEV)
    w_t = ops.prep_weights(wd32, 0, dtype)
    w_d = ops.prep_weights(wd32, 1, dtype)
    stats = ops.new_stats(Cout, DEV)
    y = ops.conv2d_fwd(xd, w_t, k, k, stride, pad, OH, OW, stats=stats)
    sums = ops.bn_reduce_slots(stats)
    dx = ops.conv2d_dgrad(dyd, w_d, k, k, stride, pad, H, W)
    dx2 = dx.clone()
    ops.conv2d_dgrad(dyd, w_d, k, k, stride, pad, H, W, out=dx2, accumulate=True)
    dw = ops.conv2d_wgrad(xd, dyd, k, k, stride, pad)
    torch.cuda.synchronize()
    tag = 'V%d %dx%d %d->%d k%d s%d %s%s' % (V, H, W, Cin, Cout, k, stride, str(dtype).split('.')[-1], '' if matmul == 'exact' else ' ' + matmul)
    b3 = matmul in ('bf16x3', 'bf16x6_3', 'f16x3_3')       # three bf16 terms in the backward GEMMs
    tf = _tol(dtype) * (2 if matmul == 'bf16x3' else 1)   # forward: 'f16x3_3' (three split-fp16 terms) is held to the exact mode's gate
    t = _tol(dtype) * (2 if b3 else 1)
    tw = (2e-5 if dtype == torch.float32 else 1e-4) * (2 if b3 else 1)
    res = [_res('conv_fwd ' + tag, y, y_ref, tf),
           _res('conv_stats_sum ' + tag, sums[0], y_ref.sum((0, 1, 2)), 1e-4, 1e-3 * float(y_ref.abs().sum((0, 1, 2)).max())),
           _res('conv_stats_sq ' + tag, sums[1], (y_ref ** 2).sum((0, 1, 2)), 1e-4),
           _res('conv_dgrad ' + tag, dx, dx_ref, t),
           _res('conv_dgrad_acc ' + tag, dx2, 2 * dx_ref, 2 * t),
           _res('conv_wgrad ' + tag, dw.view(k, k, Cin, Cout), dw_ref, tw)]
    # the error relative to the RMS of the reference (what a statistical argument predicts): reported, not gated
    for nm, a, b in (('fwd', y, y_ref), ('dgrad', dx, dx_ref), ('wgrad', dw.view(k, k, Cin, Cout), dw_ref)):
        d = (a.double().cpu() - b.double())
        res[0].setdefault('rms_rel', {})[nm] = float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
    # the one-launch pair must be bit-identical to the two single-layout copies (with and without channel padding)
    for (cip, cop) in [(0, 0), (Cin + 64, Cout + 64)]:
        pt, pd = ops.prep_weights_pair(wd32, dtype, cin_p=cip, cout_p=cop)
        rt_ = ops.prep_weights(wd32, 0, dtype, cin_p=cip, cout_p=cop)
        rd_ = ops.prep_weights(wd32, 1, dtype, cin_p=cip, cout_p=cop)
        res.append(_res('prep_pair_t pad%d ' % cip + tag, pt.float(), rt_.float(), 0.0))
        res.append(_res('prep_pair_d pad%d ' % cip + tag, pd.float(), rd_.float(), 0.0))
    return res


def check_dgrad_bn(V, H, Cin, Cout, k, dtype, mask_mode, accumulate, seed=0):
    """simclr_conv2d_dgrad_bn (dgrad + fused ReLU mask + BN-backward sums) vs float64:
    dm = (dgrad(dy) [+ prev]) * mask,  sums = (sum dm, sum dm * (x-mean)*rstd) per channel."""
    g = torch.Generator().manual_seed(seed)
    pad = (k - 1) // 2
    x_raw = _rand((V, H, H, Cin), dtype, g) * 1.5 + 0.3          # raw input of the producer BN
    w = _rand((k, k, Cin, Cout), dtype, g, (k * k * Cin) ** -0.5)
    dy = _rand((V, H, H, Cout), dtype, g)
    prev = _rand((V, H, H, Cin), dtype, g) if accumulate else None
    mask_t = _rand((V, H, H, Cin), dtype, g)                     # mode 1: sign gives the mask
    scale = torch.rand(Cin, generator=g) - 0.4
    shift = 0.3 * torch.randn(Cin, generator=g)
    mean = 0.2 * torch.randn(Cin, generator=g)
    rstd = 0.5 + torch.rand(Cin, generator=g)
    # reference
    xr = torch.zeros(V, Cin, H, H, dtype=torch.float64, requires_grad=True)
    yr = F.conv2d(F.pad(xr, (pad, k - 1 - pad, pad, k - 1 - pad)), w.double().permute(3, 2, 0, 1))
    yr.backward(dy.double().permute(0, 3, 1, 2))
    da = xr.grad.permute(0, 2, 3, 1)
    if accumulate:
        da = da + prev.double()
    if mask_mode in (1, 3, 4):
        m = mask_t.double() > 0
    else:
        m = (x_raw.float() * scale + shift).double() > 0          # fp32 fma like the kernel
    dm_ref = torch.where(m, da, torch.zeros(1, dtype=torch.float64))
    xh = (x_raw.double() - mean.double()) * rstd.double()
    s1 = dm_ref.sum((0, 1, 2)); s2 = (dm_ref * xh).sum((0, 1, 2))
    # device
    w_d = ops.prep_weights(w.float().to(DEV), 1, dtype)
    mask_arg = None
    if mask_mode == 1:
        mask_arg = mask_t.to(DEV)
    elif mask_mode in (3, 4):      # one byte per 16-byte chunk, bit e = element e (the layout simclr_bn_apply writes)
        epc = 8 if dtype == torch.bfloat16 else 4
        mb = (mask_t > 0).reshape(-1, Cin // epc, epc).to(torch.int32)
        mask_arg = (mb << torch.arange(epc, dtype=torch.int32)).sum(-1).to(torch.uint8).to(DEV)
    bn = dict(x=x_raw.to(DEV), mask=mask_arg, scale=scale.to(DEV),
              shift=shift.to(DEV), mean=mean.to(DEV), rstd=rstd.to(DEV), mode=mask_mode)
    if mask_mode == 4:        # sums-only epilogue: no BN input, no mean / rstd
        bn = dict(mask=mask_arg, mode=4)
    out = prev.to(DEV).clone() if accumulate else None
    dm, part = ops.conv2d_dgrad_bn(dy.to(DEV), w_d, k, k, pad, H, H, bn, out=out, accumulate=accumulate)
    sums = ops.bn_reduce_slots(part)
    torch.cuda.synchronize()
    tag = 'V%d %dx%d %d->%d k%d %s mode%d acc%d' % (V, H, H, Cin, Cout, k, str(dtype).split('.')[-1], mask_mode, accumulate)
    t = _tol(dtype)
    # elements whose fp32 mask argument is within rounding of 0 may legitimately flip: compare where |arg| is clear
    if mask_mode == 2:
        arg = (x_raw.float() * scale + shift).abs()
        clear = arg > 1e-3 * (arg.max() + 1e-9)
        got = torch.where(clear, dm.double().cpu(), dm_ref)
    else:
        got = dm
    res = [_res('dgrad_bn_dm ' + tag, got, dm_ref, t * (2 if accumulate else 1)),
           _res('dgrad_bn_sum ' + tag, sums[0], s1, 2e-3 if dtype == torch.bfloat16 else 1e-4, 1e-3 * float(dm_ref.abs().sum((0, 1, 2)).max()))]
    if mask_mode != 4:
        res.append(_res('dgrad_bn_sumxhat ' + tag, sums[1], s2, 2e-3 if dtype == torch.bfloat16 else 1e-4, 1e-3 * float((dm_ref * xh).abs().sum((0, 1, 2)).max())))
    return res


def check_stem(V, H, k, stride, Cout, dtype, seed=0, matmul='exact'):
    """matmul (fp32 only): simclr_set_f32_matmul mode -- the split-bf16 stem forward (stem_conv_fwd<float, ., 0, 3 | 6>) and the LDS-DMA
    stem weight gradient (conv_wgrad_dma<float, 256, 64, ..., MT, 3 | 6>) against the same float64 reference."""
    ops.set_f32_matmul(matmul if dtype == torch.float32 else 'exact')
    try:
        return _check_stem(V, H, k, stride, Cout, dtype, seed, matmul)
    finally:
        ops.set_f32_matmul('exact')


def _check_stem(V, H, k, stride, Cout, dtype, seed, matmul):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(V // 2, H, H, 6, generator=g)
    w = _rand((k, k, 3, Cout), dtype, g, (k * k * 3) ** -0.5)
    geo = ops.stem_geometry(H, H, k, k, stride)
    dy = _rand((V, geo['OH'], geo['OW'], Cout), dtype, g)
    views = torch.cat(torch.split(img, 3, dim=3), 0).to(dtype)      # tf2/model.py:250-259
    xr = views.double().permute(0, 3, 1, 2)
    wr = w.double().requires_grad_(True)
    pad = geo['pad']
    pe = (k - 1) - pad
    yr = F.conv2d(F.pad(xr, (pad, pe, pad, pe)), wr.permute(3, 2, 0, 1), stride=stride)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    xp = ops.pack_views(img.to(DEV), 2, geo, dtype)
    w_s = ops.prep_weights(w.float().to(DEV), 2, dtype, geo['KHP'], geo['KWP'])
    stats = ops.new_stats(Cout, DEV)
    y = ops.stem_conv_fwd(xp, w_s, geo, stride, stats=stats)
    sums = ops.bn_reduce_slots(stats)
    dw = ops.stem_conv_wgrad(xp, dy.to(DEV), geo, k, k, stride)
    torch.cuda.synchronize()
    tag = 'V%d %d k%d s%d ->%d %s%s' % (V, H, k, stride, Cout, str(dtype).split('.')[-1], '' if matmul == 'exact' else '/' + matmul)
    t = _tol(dtype)
    y_ref = yr.detach().permute(0, 2, 3, 1)
    bwd_scale = 2.0 if matmul in ('bf16x3', 'bf16x6_3', 'f16x3_3') else 1.0          # three-term backward arithmetic: ~2^-17 per product
    fwd_scale = 4.0 if matmul == 'bf16x3' else 1.0
    res = [_res('stem_fwd ' + tag, y, y_ref, t * fwd_scale),
           _res('stem_stats_sq ' + tag, sums[1], (y_ref ** 2).sum((0, 1, 2)), 1e-4),
           _res('stem_wgrad ' + tag, dw, wr.grad, (2e-5 if dtype == torch.float32 else 1e-4) * bwd_scale)]
    if dtype == torch.float32 and ops.stem_wgrad_ps_supported(geo, k, stride, Cout):
        # the same weight gradient from pre-split operands (simclr_stem_wgrad_ps: image pieces per packed pixel, gradient in the block
        # format): the pieces are the three-term operands, so the float64 bar of the in-register split applies
        dw_ps = ops.stem_conv_wgrad(xp, ps_encode(dy.to(DEV)), geo, k, k, stride)
        xq = ops.presplit_packed(xp)
        xp2, xq2 = ops.pack_views(img.to(DEV), 2, geo, dtype, with_presplit=True)      # the same two tensors from ONE pass over the images
        raw = xq.view(torch.int16).view(torch.bfloat16).reshape(-1, 8).float().cpu()
        xf = xp.reshape(-1, 4).cpu()
        torch.cuda.synchronize()
        res += [_res('stem_wgrad_presplit ' + tag, dw_ps, wr.grad, 2e-5 * bwd_scale),
                _res('stem_xq_hi ' + tag, raw[:, :4], xf.bfloat16().float(), 0.0),
                _res('stem_xq_lo ' + tag, raw[:, 4:], (xf - xf.bfloat16().float()).bfloat16().float(), 0.0),
                _res('stem_pack_ps_xp ' + tag, xp2, xp, 0.0),
                _res('stem_pack_ps_xq ' + tag, xq2.view(torch.int32).double(), xq.view(torch.int32).double(), 0.0)]
    return res


# ------------------------------------------------------------------ BN
def check_bn(rows_shape, C, dtype, relu, residual, seed=0):
    """residual: None | 'identity' | 'bn'"""
    g = torch.Generator().manual_seed(seed)
    shape = tuple(rows_shape) + (C,)
    x = _rand(shape, dtype, g) * 2 + 0.5
    gamma = (0.5 + torch.rand(C, generator=g))
    beta = 0.1 * torch.randn(C, generator=g)
    dy = _rand(shape, dtype, g)
    res_t = _rand(shape, dtype, g) if residual else None
    mm, mv = torch.zeros(C), torch.ones(C)
    # reference (float64)
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    axes = tuple(range(len(shape) - 1))
    mean = xr.mean(axes)
    var = ((xr - mean) ** 2).mean(axes)
    y = (xr - mean) * torch.rsqrt(var + 1e-5) * gr + br
    rr = None
    if residual == 'identity':
        rr = res_t.double().requires_grad_(True)
        y = y + rr
    elif residual == 'bn':
        rr = res_t.double().requires_grad_(True)
        y = y + (rr * 1.5 - 0.25)
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    # device
    xd = x.to(DEV)
    m = x.numel() // C
    xf = xd.float().view(-1, C)
    part = ops.new_stats(C, DEV)
    part[0, 0] = xf.double().sum(0).float()          # statistics normally come from the conv epilogue
    part[0, 1] = (xf.double() ** 2).sum(0).float()
    sums = ops.bn_reduce_slots(part)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    mmd, mvd = mm.to(DEV), mv.to(DEV)
    mean_d, rstd_d, scale, shift = ops.bn_finalize(sums, m, gd, bd, mmd, mvd, 0.9)
    rs = torch.full((C,), 1.5, device=DEV) if residual == 'bn' else None
    rb = torch.full((C,), -0.25, device=DEV) if residual == 'bn' else None
    yd, bits = ops.bn_apply(xd, scale, shift, relu, res=res_t.to(DEV) if residual else None, rscale=rs, rshift=rb,
                            want_bits=True)
    mask_mode = 1 if relu else 0
    dyd = dy.to(DEV)
    p = ops.bn_bwd_reduce(dyd, xd, yd, scale, shift, mean_d, rstd_d, mask_mode)
    p2 = ops.bn_bwd_reduce(dyd, xd, None, scale, shift, mean_d, rstd_d, 2 if (relu and not residual) else mask_mode) if not residual else None
    ls = ops.bn_reduce_slots(p)
    dgamma = torch.zeros(C, device=DEV); dbeta = torch.zeros(C, device=DEV)
    c1, c2 = ops.bn_bwd_finalize(ls, ls, m, dgamma, dbeta)
    dx, dmask = ops.bn_bwd_apply(dyd, xd, yd, scale, shift, mean_d, rstd_d, c1, c2, mask_mode, want_masked=True)
    torch.cuda.synchronize()
    tag = '%s C%d %s relu=%d res=%s' % ('x'.join(map(str, rows_shape)), C, str(dtype).split('.')[-1], relu, residual)
    t = _tol(dtype)
    out = [_res('bn_apply ' + tag, yd, y.detach(), t, 1e-6),
           _res('bn_moving_mean ' + tag, mmd, 0.1 * mean.detach(), 1e-5, 1e-7),
           _res('bn_moving_var ' + tag, mvd, 0.9 + 0.1 * var.detach(), 1e-5),
           _res('bn_dgamma ' + tag, dgamma, gr.grad, 2e-4 if dtype == torch.float32 else 2e-3),
           _res('bn_dbeta ' + tag, dbeta, br.grad, 2e-4 if dtype == torch.float32 else 2e-3),
           _res('bn_dx ' + tag, dx, xr.grad, 5e-5 if dtype == torch.float32 else 1e-2)]
    # the ReLU bit tensor must be exactly (stored y > 0), one byte per 16-byte chunk, LSB = first element
    epc = 8 if dtype == torch.bfloat16 else 4
    yb = (yd.reshape(-1, C // epc, epc) > 0).to(torch.int32)
    want_bits = (yb << torch.arange(epc, dtype=torch.int32, device=DEV)).sum(-1).to(torch.uint8)
    out.append(_res('bn_relu_bits ' + tag, bits.reshape(-1, C // epc).float(), want_bits.float(), 0.0, 0.0))
    if residual == 'identity':
        out.append(_res('bn_dmasked ' + tag, dmask, rr.grad, t))
    if p2 is not None and relu:
        # mask recomputed from x*scale+shift (mask_mode 2) must give the same sums as mask from y
        out.append(_res('bn_bwd_reduce_mode2 ' + tag, ops.bn_reduce_slots(p2), ls, 1e-6, 1e-6))
        dx2, _ = ops.bn_bwd_apply(dyd, xd, None, scale, shift, mean_d, rstd_d, c1, c2, 2)
        out.append(_res('bn_dx_mode2 ' + tag, dx2, xr.grad, 5e-5 if dtype == torch.float32 else 1e-2))
    return out


def check_pool(V, H, C, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = _rand((V, H, H, C), dtype, g)
    scale = (torch.rand(C, generator=g) - 0.3)          # some negative scales on purpose
    shift = 0.2 * torch.randn(C, generator=g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    a = F.relu(xr * scale.double().view(1, C, 1, 1) + shift.double().view(1, C, 1, 1))
    OH, pt = ops.same_pad(H, 3, 2)
    total = max((OH - 1) * 2 + 3 - H, 0)
    pr = F.max_pool2d(F.pad(a, (pt, total - pt, pt, total - pt), value=float('-inf')), 3, 2)
    dy = _rand((V, OH, OH, C), dtype, g)
    pr.backward(dy.double().permute(0, 3, 1, 2))
    d_act_ref = (xr.grad / torch.where(scale == 0, torch.ones_like(scale), scale).double().view(1, C, 1, 1))
    y, arg = ops.bnrelu_maxpool_fwd(x.to(DEV), scale.to(DEV), shift.to(DEV))
    da = ops.maxpool_bwd(dy.to(DEV), arg, H, H)
    # avg pool
    p = ops.global_avgpool_fwd(x.to(DEV))
    dp = _rand((V, C), dtype, g)
    dxa = ops.global_avgpool_bwd(dp.to(DEV), H, H)
    torch.cuda.synchronize()
    tag = 'V%d %d C%d %s' % (V, H, C, str(dtype).split('.')[-1])
    t = _tol(dtype)
    # d_act: gradient wrt the ReLU output routed to the argmax; compare where the ReLU is active
    act = (a.detach() > 0).permute(0, 2, 3, 1)
    da_ref = torch.where(act, d_act_ref.permute(0, 2, 3, 1), torch.zeros(1, dtype=torch.float64))
    da_got = torch.where(act, da.double().cpu(), torch.zeros(1, dtype=torch.float64))
    return [_res('maxpool_fwd ' + tag, y, pr.detach().permute(0, 2, 3, 1), t, 1e-6),
            _res('maxpool_bwd ' + tag, da_got, da_ref, t, 1e-6),
            _res('avgpool_fwd ' + tag, p, x.double().mean((1, 2)), t, 1e-6),
            _res('avgpool_bwd ' + tag, dxa, (dp.double() / (H * H))[:, None, None, :].expand(V, H, H, C), t, 1e-7)]


def check_sup_head(rows, nclass, cpad, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    z = _rand((rows, cpad), dtype, g) * 3
    bias = 0.1 * torch.randn(nclass, generator=g)
    labels = torch.randint(0, nclass, (rows // 2,), generator=g).int()
    lab2 = torch.cat([labels, labels]).long()
    zr = (z.double()[:, :nclass] + bias.double()).requires_grad_(True)
    loss = F.cross_entropy(zr, lab2)
    loss.backward()
    out = torch.zeros(2, device=DEV)
    dl = ops.bias_softmax_xent(z.to(DEV), bias.to(DEV), labels.to(DEV), nclass, 1.0, out)
    db = torch.zeros(nclass, device=DEV)
    ops.colsum(dl, nclass, db)
    torch.cuda.synchronize()
    acc = (zr.argmax(1) == lab2).double().mean()
    tag = '%dx%d(%d) %s' % (rows, nclass, cpad, str(dtype).split('.')[-1])
    t = _tol(dtype)
    return [_res('sup_loss ' + tag, out[0], loss.detach(), 1e-5),
            _res('sup_acc ' + tag, out[1], acc, 0, 1e-6),
            _res('sup_dlogits ' + tag, dl[:, :nclass], zr.grad, t, 1e-7),
            _res('sup_dlogits_pad ' + tag, dl[:, nclass:], torch.zeros(rows, cpad - nclass), 0, 0),
            _res('sup_dbias ' + tag, db, zr.grad.sum(0), 10 * t, 1e-6)]


# ------------------------------------------------------------------ end-to-end training step
def _flat(d, keys, like):
    return torch.cat([(d[k] if d[k] is not None else torch.zeros_like(like[k])).double().reshape(-1).cpu() for k in keys])


def check_blur(b, H, k=2, seed=0):
    """simclr_batch_blur vs the oracle restatement of tf2/data_util.py gaussian_blur / batch_random_blur."""
    from oracle import blur as oblur
    from simclr_amd import data_util
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, H, H, 3 * k, generator=g)
    sigmas = [0.1 + 1.9 * float(torch.rand(1, generator=g)) for _ in range(k)]
    sel = (torch.rand(k, b, generator=g) < 0.5).float()
    ref = oblur.batch_random_blur([x[..., 3 * i:3 * i + 3].numpy() for i in range(k)], H, sigmas, sel.numpy())
    ref = np.concatenate(ref, axis=3)
    y = data_util.batch_random_blur_tensor(x.to(DEV), H, H, sigmas=sigmas, selectors=sel)
    torch.cuda.synchronize()
    return [_res('batch_blur b%d %dpx k%d' % (b, H, k), y, ref, 0, 2e-6)]


def check_avgpool2(V, H, C, stride, dtype, seed=0):
    """ResNet-D shortcut pool vs the oracle's restatement of AveragePooling2D (tf2/resnet.py:330-338)."""
    from oracle.model_torch import Builder, Config
    g = torch.Generator().manual_seed(seed)
    x = _rand((V, H, H, C), dtype, g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = Builder(Config(sk_ratio=0.0625)).avg_pool(xr, stride)
    dy = _rand(tuple(yr.permute(0, 2, 3, 1).shape), dtype, g)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    y = ops.avgpool2_fwd(x.to(DEV), stride)
    dx = ops.avgpool2_bwd(dy.to(DEV), H, H, stride)
    torch.cuda.synchronize()
    tag = 'V%d %d C%d s%d %s' % (V, H, C, stride, str(dtype).split('.')[-1])
    t = _tol(dtype)
    return [_res('avgpool2_fwd ' + tag, y, yr.detach().permute(0, 2, 3, 1), t, 1e-6),
            _res('avgpool2_bwd ' + tag, dx, xr.grad.permute(0, 2, 3, 1), t, 1e-6)]


_TRAIN_STEP_ORACLE_CACHE = {}


def check_train_step(depth=18, image_size=32, batch=8, compute_dtype='f32', num_classes=10, seed=0,
                     weight_decay=1e-4, lr=0.1, steps=1, randomize_bn=True, sk_ratio=0.0, width_multiplier=1,
                     proj_out_dim=128, inputs='iid'):
    """Full pretraining steps: HIP path vs the torch-CPU oracle restating tf2/run.py:557-622 on
    identical weights and inputs.

    Ground truth is the oracle in float64.  ReLU sign flips and small-batch BatchNorm make the step
    ill-conditioned, so the tolerance of every quantity is CALIBRATED by the error the reference
    arithmetic itself shows at the same precision: the oracle run in float32 (f32 mode) or with
    bf16 rounding of weights/activations emulated (bf16 mode), each against float64:
        tol = CAL * err(oracle@precision vs f64) + floor.
    Steps after the first are teacher-forced (weights/momenta re-synchronised from the oracle) so
    each step is an independent parity case that also exercises the weight-refresh logic."""
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, train_step
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step

    CAL = 6.0     # our fp32 kernels vs float64 may deviate a few times more than torch-CPU fp32 does (different fusion / summation order)
    okey = (depth, image_size, batch, compute_dtype == 'bf16', num_classes, seed, weight_decay, lr, randomize_bn, sk_ratio, width_multiplier,
            proj_out_dim, inputs) if steps == 1 else None         # the oracle's two steps of this configuration (several tests share one)
    cfg = Config(resnet_depth=depth, image_size=image_size, num_classes=num_classes, weight_decay=weight_decay,
                 sk_ratio=sk_ratio, width_multiplier=width_multiplier, proj_out_dim=proj_out_dim)
    params, state = init_model(cfg, seed=seed, randomize_bn=randomize_bn)
    momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
    g = torch.Generator().manual_seed(seed + 1)
    FLAGS.reset()
    FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype=compute_dtype, use_blur=False,
                 weight_decay=weight_decay, train_batch_size=batch, sk_ratio=sk_ratio,
                 width_multiplier=width_multiplier, proj_out_dim=proj_out_dim)
    RT.reset()
    RT.device = torch.device(DEV)
    model = model_lib.Model(num_classes)
    with torch.no_grad():
        model(torch.zeros(2, image_size, image_size, 6, device=DEV), training=True)   # builds the variables
    res = []
    names_model = [v.name for v in model.variables]
    names_oracle = list(params.keys()) + list(state.keys())
    missing = sorted(set(names_oracle) - set(names_model))
    extra = sorted(set(names_model) - set(names_oracle))
    tag = 'R%d%s%s %dpx b%d %s%s' % (depth, '' if width_multiplier == 1 else ' %dx' % width_multiplier,
                                     ' SK' if sk_ratio > 0 else '', image_size, batch, compute_dtype,
                                     '' if randomize_bn else ' refinit')
    res.append(dict(name='step_var_names ' + tag, err=float(len(missing) + len(extra)), tol=0.0, scale=0.0,
                    ok=not missing and not extra, nbad=len(missing) + len(extra), numel=len(names_oracle),
                    missing=missing[:5], extra=extra[:5]))
    if missing or extra:
        return res
    optimizer = model_lib.build_optimizer(lr)
    step_fn = make_single_step(model, optimizer, None)
    keys = list(params.keys())
    emu = compute_dtype == 'bf16'

    def entry(name, err, ref_err, floor, cal=CAL, cap=None, **kw):
        # calibrated tolerance, but never vacuous: `cap` is a fixed upper bound on the gate (gates whose
        # calibrated tol exceeded 1.0 said nothing); the realistic-batch fixed-threshold gates are in check_train_step_fixed
        tol = cal * ref_err + floor
        if cap is not None:
            tol = min(tol, cap)
        d = dict(name=name, err=float(err), tol=float(tol), scale=float(ref_err), ok=bool(err <= tol), nbad=0, numel=1)
        d.update(kw)
        return d

    for s_i in range(steps):
        # (re)load weights, BN state and momenta from the oracle
        allv = dict(params); allv.update(state)
        for v in model.variables:
            assert tuple(v.value.shape) == tuple(allv[v.name].shape), (v.name, v.value.shape, allv[v.name].shape)
            v.value.copy_(allv[v.name].to(DEV))
        if s_i > 0:
            for v in model._flat_order:
                optimizer.get_slot(v, 'Momentum').copy_(momenta[v.name].to(DEV))
        RT.weights_version += 1
        # 'structured': image-like inputs (check_train_step_fixed) -- i.i.d. noise images are statistically identical, and a
        # deep untrained network maps them to nearly the same feature: every BatchNorm then normalises a mean hundreds of
        # standard deviations away from zero, which no fp32 sum-of-squares statistic survives
        images = (torch.rand(batch, image_size, image_size, 6, generator=g) if inputs == 'iid'
                  else structured_images(batch, image_size, 2, g))
        labels = torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float()
        p64 = OrderedDict((k, v.double()) for k, v in params.items())
        s64 = OrderedDict((k, v.double()) for k, v in state.items())
        m64 = OrderedDict((k, v.double()) for k, v in momenta.items())
        if okey is not None and okey in _TRAIN_STEP_ORACLE_CACHE:
            (np64, ns64, nm64, t64), (np32, ns32, nm32, t32) = _TRAIN_STEP_ORACLE_CACHE[okey]
        else:
            np64, ns64, nm64, t64 = train_step(cfg, p64, s64, m64, images.double(), labels.double(), lr)
            np32, ns32, nm32, t32 = train_step(cfg, params, state, momenta, images, labels, lr, emulate_bf16=emu)
            if okey is not None and sum(v.numel() for v in params.values()) < 60e6:       # small models only: the cache holds 4 copies
                _TRAIN_STEP_ORACLE_CACHE.clear()
                _TRAIN_STEP_ORACLE_CACHE[okey] = ((np64, ns64, nm64, t64), (np32, ns32, nm32, t32))
        out = step_fn(images.to(DEV), {'labels': labels.to(DEV)})
        torch.cuda.synchronize()
        st = ' step%d' % s_i

        def rel(a, b):
            return float((a.double().cpu() - b.double()).abs().max()) / (float(b.double().abs().max()) + 1e-30)

        for nm_, mine, k in (('con_loss', out['con_loss'].value, 'con_loss'), ('sup_loss', out['sup_loss'].value, 'sup_loss'),
                             ('total_loss', out['total_loss'], 'total_loss')):
            res.append(entry('step_%s %s%s' % (nm_, tag, st), rel(mine.reshape(-1)[0], t64[k].detach()),
                             rel(t32[k].detach(), t64[k].detach()), 1e-5 if not emu else 1e-2,
                             value=float(mine.reshape(-1)[0]), ref=float(t64[k].detach())))
        z_err = float((out['con_loss'].normalized.double().cpu() - t64['z'].detach()).abs().max())
        z_ref = float((t32['z'].detach().double() - t64['z'].detach()).abs().max())
        res.append(entry('step_embeddings_abs %s%s' % (tag, st), z_err, z_ref, 2e-6 if not emu else 1e-2))
        g64 = _flat(t64['grads'], keys, p64)
        g32 = _flat(t32['grads'], keys, params)
        gm = torch.cat([{v.name: v for v in model._flat_order}[k].grad.double().reshape(-1).cpu() for k in keys])
        cos_m = float((gm * g64).sum() / gm.norm() / g64.norm())
        cos_r = float((g32 * g64).sum() / g32.norm() / g64.norm())
        res.append(entry('step_grad_1-cos %s%s' % (tag, st), 1 - cos_m, 1 - cos_r, 2e-4 if not emu else 2e-2, cal=8.0,
                         cap=5e-3 if not emu else 0.2))
        res.append(entry('step_grad_relnorm %s%s' % (tag, st), float((gm - g64).norm() / g64.norm()),
                         float((g32 - g64).norm() / g64.norm()), 2e-2 if not emu else 1e-1, cap=0.1 if not emu else 0.7))
        # Per-tensor relative errors.  A handful of ReLU pre-activations per step lie within fp32
        # rounding noise of zero; which side they fall on differs between ANY two fp32 evaluations
        # (ours is not even run-to-run deterministic: statistic atomics), and one flipped element is a
        # visible fraction of a tiny late-layer gradient.  So the gate is on robust statistics (median and
        # 90th percentile over tensors, calibrated by the oracle's own fp32-vs-fp64 errors); the worst
        # tensor is reported and only loosely bounded.
        byname = {v.name: v for v in model._flat_order}
        em_l, er_l, wn, worst_m = [], [], '', 0.0
        for k in keys:
            ref = t64['grads'][k]
            if ref is None or float(ref.abs().max()) < 1e-12:
                continue
            em, er = rel(byname[k].grad, ref), rel(t32['grads'][k], ref)
            em_l.append(em); er_l.append(er)
            if em > worst_m:
                worst_m, wn = em, k
        em_t, er_t = torch.tensor(em_l), torch.tensor(er_l)
        res.append(entry('step_grad_tensor_rel_median %s%s' % (tag, st), float(em_t.median()), float(er_t.median()),
                         2e-5 if not emu else 5e-2, cal=8.0, cap=0.2 if not emu else 0.7))
        res.append(entry('step_grad_tensor_rel_p90 %s%s' % (tag, st), float(em_t.quantile(0.9)), float(er_t.quantile(0.9)),
                         1e-4 if not emu else 1e-1, cal=6.0, cap=0.4 if not emu else 0.9))
        # worst single tensor, measured against the GLOBAL gradient norm (an absolute per-tensor bound: one layer being
        # off by 30 % of its own norm shows up here unless that layer's gradient is negligible for the update)
        gn64 = float(g64.norm())
        wt_m = max(float((byname[k].grad.double().cpu() - t64['grads'][k]).norm()) / gn64 for k in keys if t64['grads'][k] is not None)
        wt_r = max(float((t32['grads'][k].double() - t64['grads'][k]).norm()) / gn64 for k in keys if t64['grads'][k] is not None)
        res.append(entry('step_grad_worst_tensor_vs_global_norm %s%s' % (tag, st), wt_m, wt_r,
                         1e-3 if not emu else 5e-2, cal=8.0, cap=0.1 if not emu else 0.5, worst=wn, worst_rel=worst_m))
        pm_l = torch.tensor([rel(byname[k].value, np64[k]) for k in keys])
        pr_l = torch.tensor([rel(np32[k], np64[k]) for k in keys])
        res.append(entry('step_new_params_rel_median %s%s' % (tag, st), float(pm_l.median()), float(pr_l.median()),
                         2e-5 if not emu else 1e-2, cal=8.0))
        res.append(entry('step_new_params_rel_p90 %s%s' % (tag, st), float(pm_l.quantile(0.9)), float(pr_l.quantile(0.9)),
                         1e-5 if not emu else 1e-1, cal=6.0))
        mk = [k for k in keys if float(nm64[k].abs().max()) > 1e-12]
        mv_l = torch.tensor([rel(optimizer.get_slot(byname[k], 'Momentum'), nm64[k]) for k in mk])
        mr_l = torch.tensor([rel(nm32[k], nm64[k]) for k in mk])
        res.append(entry('step_momentum_rel_median %s%s' % (tag, st), float(mv_l.median()), float(mr_l.median()),
                         2e-5 if not emu else 5e-2, cal=8.0, cap=0.2 if not emu else 0.7))
        bm = max(rel(v.value, ns64[v.name]) for v in model.variables if v.name in ns64)
        br = max(rel(ns32[k], ns64[k]) for k in ns64)
        res.append(entry('step_bn_moving_worst_rel %s%s' % (tag, st), bm, br, 1e-5 if not emu else 1e-2))
        params = OrderedDict((k, v.float()) for k, v in np64.items())
        state = OrderedDict((k, v.float()) for k, v in ns64.items())
        momenta = OrderedDict((k, v.float()) for k, v in nm64.items())
    return res


def check_eval_and_checkpoint(depth=18, image_size=32, batch=8, num_classes=10, seed=0):
    """(f)-3: eval-mode forward (BN on moving statistics, tf2/resnet.py:62-72 with training=False) vs the oracle,
    and checkpoint -> fresh model -> restore -> identical continuation (tf2/run.py:308-337), then
    perform_evaluation (tf2/run.py:348-432) on that checkpoint."""
    import json
    import os
    import tempfile
    from oracle.model_torch import Builder, Config, init_model
    from simclr_amd import model as model_lib
    from simclr_amd.checkpoint import Checkpoint, CheckpointManager, try_restore_from_checkpoint
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step, perform_evaluation, synthetic_eval_batches

    cfg = Config(resnet_depth=depth, image_size=image_size, num_classes=num_classes)
    params, state = init_model(cfg, seed=seed, randomize_bn=True)
    g = torch.Generator().manual_seed(seed + 7)

    def fresh_model():
        FLAGS.reset()
        FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype='f32', use_blur=False,
                     train_batch_size=batch)
        RT.reset()
        RT.device = torch.device(DEV)
        m = model_lib.Model(num_classes)
        with torch.no_grad():
            m(torch.zeros(2, image_size, image_size, 6, device=DEV), training=True)     # builds the variables
        return m

    res = []
    model = fresh_model()
    allv = dict(params); allv.update(state)
    for v in model.variables:
        v.value.copy_(allv[v.name].to(DEV))
    RT.weights_version += 1
    # ---- eval forward vs oracle (float64)
    x = torch.rand(batch, image_size, image_size, 3, generator=g)
    from collections import OrderedDict
    b64 = Builder(cfg, params=OrderedDict((k, v.double()) for k, v in params.items()),
                  state=OrderedDict((k, v.double()) for k, v in state.items()), dtype=torch.float64)
    with torch.no_grad():
        proj_ref, sup_ref = b64.model(x.double(), training=False)
    proj, sup = model(x.to(DEV), training=False)
    torch.cuda.synchronize()
    tag = 'R%d %dpx b%d f32' % (depth, image_size, batch)
    res.append(_res('eval_proj ' + tag, proj, proj_ref, 2e-4, 1e-6))
    res.append(_res('eval_sup_logits ' + tag, sup.dense(), sup_ref, 2e-4, 1e-6))
    # moving statistics must NOT move in eval mode
    mm_now = {v.name: v.value.clone() for v in model.variables if 'moving_' in v.name}
    drift = max(float((mm_now[k].cpu() - allv[k]).abs().max()) for k in mm_now)
    res.append(dict(name='eval_moving_stats_untouched ' + tag, err=drift, tol=0.0, scale=0.0, ok=drift == 0.0, nbad=0, numel=len(mm_now)))

    # ---- train 2 steps, checkpoint, one more step = the continuation to reproduce
    optimizer = model_lib.build_optimizer(0.1)
    step_fn = make_single_step(model, optimizer, None)
    feats = [torch.rand(batch, image_size, image_size, 6, generator=g).to(DEV) for _ in range(3)]
    labs = [{'labels': torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float().to(DEV)}
            for _ in range(3)]
    for i in range(2):
        step_fn(feats[i], labs[i])
    d = tempfile.mkdtemp(prefix='simclr_ckpt_')
    mgr = CheckpointManager(Checkpoint(model=model, optimizer=optimizer), d, max_to_keep=5)
    path = mgr.save()
    out3 = step_fn(feats[2], labs[2])
    torch.cuda.synchronize()
    l3 = float(out3['total_loss'].reshape(-1)[0])
    w3 = {v.name: v.value.clone() for v in model.variables}

    model2 = fresh_model()
    opt2 = model_lib.build_optimizer(0.1)
    step2 = make_single_step(model2, opt2, None)
    mgr2, status = try_restore_from_checkpoint(model2, opt2, d)
    unused = len(status.missing_in_checkpoint) + len(status.unused_in_checkpoint)
    res.append(dict(name='ckpt_fully_consumed ' + tag, err=float(unused), tol=0.0, scale=0.0, ok=unused == 0 and opt2.iterations == 2,
                    nbad=unused, numel=len(w3), path=os.path.basename(path)))
    o3 = step2(feats[2], labs[2])
    torch.cuda.synchronize()
    l3b = float(o3['total_loss'].reshape(-1)[0])
    res.append(dict(name='ckpt_resume_loss ' + tag, err=abs(l3b - l3) / abs(l3), tol=2e-5, scale=abs(l3), ok=abs(l3b - l3) / abs(l3) <= 2e-5,
                    nbad=0, numel=1, value=l3b, ref=l3))
    num = den = 0.0
    werr = 0.0
    for v in model2.variables:
        ref = w3[v.name]
        num += float(((v.value - ref).double() ** 2).sum()); den += float((ref.double() ** 2).sum())
        werr = max(werr, float((v.value - ref).abs().max()) / (float(ref.abs().max()) + 1e-12))
    gerr = (num / den) ** 0.5
    # BatchNorm statistics, LARS norms and every other reduction on the weight path use fixed summation orders
    # (one slot per workgroup, ordered slot sums): the continuation is reproduced exactly
    res.append(dict(name='ckpt_resume_weights_rel_l2 ' + tag, err=gerr, tol=1e-7, scale=1.0, ok=gerr <= 1e-7, nbad=0, numel=len(w3)))
    res.append(dict(name='ckpt_resume_weights_worst_rel ' + tag, err=werr, tol=1e-6, scale=1.0, ok=werr <= 1e-6, nbad=0, numel=len(w3)))

    # ---- perform_evaluation on that checkpoint with a third, untouched model
    model3 = fresh_model()
    data = synthetic_eval_batches(batch, image_size, num_classes, torch.device(DEV), seed=3)
    result = perform_evaluation(model3, data, 2, path, None, model_dir=d)
    files_ok = all(os.path.exists(os.path.join(d, f)) for f in ('result.json', 'result_2.json', 'flags.json'))
    on_disk = json.load(open(os.path.join(d, 'result.json'))) if files_ok else {}
    ok = (files_ok and result['global_step'] == 2 and on_disk.get('global_step') == 2.0 and
          0.0 <= result['eval/label_top_1_accuracy'] <= result['eval/label_top_5_accuracy'] <= 1.0 and
          result['eval/regularization_loss'] > 0.0)
    res.append(dict(name='perform_evaluation ' + tag, err=0.0 if ok else 1.0, tol=0.0, scale=0.0, ok=bool(ok), nbad=0, numel=1,
                    result=result))
    # the restored model3 must give the same logits as model (weights at step 2 were overwritten by step 3 in `model`,
    # so compare against model2 rolled back: reload the checkpoint into model2 and evaluate both on one batch)
    Checkpoint(model=model2).restore(path, model_only=True)
    xe = torch.rand(batch, image_size, image_size, 3, generator=g).to(DEV)
    _, s2 = model2(xe, training=False)
    _, s3 = model3(xe, training=False)
    torch.cuda.synchronize()
    res.append(_res('eval_after_restore_identical ' + tag, s3.dense(), s2.dense(), 1e-6, 1e-7))
    return res


# ------------------------------------------------------------------ conv at the sizes bench.py runs
def _guarded(shape, dtype, fill=None):
    """A tensor with sentinel-filled guard bands on either side (one allocation).  Returns (view, check)
    where check() -> number of guard elements a kernel overwrote (out-of-bounds WRITES)."""
    n = 1
    for s in shape:
        n *= s
    band = 4096
    buf = torch.empty(n + 2 * band, device=DEV, dtype=dtype)
    sent = 1234.5 if dtype != torch.uint8 else 77
    buf[:band] = sent
    buf[band + n:] = sent
    view = buf[band:band + n].view(*shape)
    if fill is not None:
        view.copy_(fill)

    def check():
        return int((buf[:band] != sent).sum()) + int((buf[band + n:] != sent).sum())
    return view, check


def _ref64_conv(x, w, dy, k, s, pad, OH, OW, vchunk):
    """Plain-torch float64 reference on the device, chunked over views so the float64 temporaries stay small:
    y = sum_taps shift(x) @ w[tap] (Conv2dFixedPadding, tf2/resnet.py:183-208), dx / dw = its exact adjoints.
    Yields per chunk (v0, v1, y64, dx64) and accumulates dw64, sum(y), sum(y^2)."""
    V, H, W, Cin = x.shape
    Cout = w.shape[3]
    w64 = w.double()
    dw = torch.zeros(k, k, Cin, Cout, device=x.device, dtype=torch.float64)
    pe = (k - 1) - pad
    for v0 in range(0, V, vchunk):
        v1 = min(V, v0 + vchunk)
        xp = F.pad(x[v0:v1].double(), (0, 0, pad, pe, pad, pe))
        dyc = dy[v0:v1].double()
        y = torch.zeros(v1 - v0, OH, OW, Cout, device=x.device, dtype=torch.float64)
        dxp = torch.zeros_like(xp)
        for ty in range(k):
            for tx in range(k):
                xs = xp[:, ty:ty + s * (OH - 1) + 1:s, tx:tx + s * (OW - 1) + 1:s, :]
                y += xs @ w64[ty, tx]
                dxp[:, ty:ty + s * (OH - 1) + 1:s, tx:tx + s * (OW - 1) + 1:s, :] += dyc @ w64[ty, tx].t()
                dw[ty, tx] += xs.reshape(-1, Cin).t() @ dyc.reshape(-1, Cout)
        dx = dxp[:, pad:pad + H, pad:pad + W, :]
        yield v0, v1, y, dx, dw


def ps_decode(t, kind='b16'):
    """Pre-split block format (simclr_amd/csrc/common.h) -> float64: per 128-byte block of 32 channels, 16-byte chunk g < 4 holds the hi
    pieces of channels {4g..4g+3, 16+4g..16+4g+3}, chunk 4 + g their lo pieces; value = hi + lo.  Returns (value, hi, lo)."""
    C = t.shape[-1]
    raw = t.detach().contiguous().view(torch.int16).reshape(-1, C // 32, 8, 8).cpu()
    pt = torch.bfloat16 if kind == 'b16' else torch.float16
    pieces = raw.view(pt).double()                              # [rows, blocks, chunk, 8]
    ch = torch.tensor([[4 * g + i if i < 4 else 16 + 4 * g + i - 4 for i in range(8)] for g in range(4)])      # [4, 8] -> channel in block
    hi = torch.zeros(raw.shape[0], C // 32, 32, dtype=torch.float64)
    lo = torch.zeros_like(hi)
    hi[:, :, ch.reshape(-1)] = pieces[:, :, :4].reshape(raw.shape[0], C // 32, 32)
    lo[:, :, ch.reshape(-1)] = pieces[:, :, 4:].reshape(raw.shape[0], C // 32, 32)
    shp = tuple(t.shape)
    return (hi + lo).reshape(shp), hi.reshape(shp), lo.reshape(shp)


def check_ps_weights(V, H, Cin, Cout, k, seed=0):
    """Pre-split weight copies made once per refresh (ops.WeightPairBatch -> simclr_presplit_weights_multi, SIMCLR_FMT_PS_W) against the copies
    the library makes per launch: same pieces, same kernels -- forward (with pivoted statistics), fused forward tail and data gradient must
    agree BIT FOR BIT."""
    ops.set_f32_matmul('f16x3_3')
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        pad = (k - 1) // 2
        w = torch.randn(k, k, Cin, Cout, device=DEV, generator=g) * (k * k * Cin) ** -0.5
        batch = ops.WeightPairBatch([(w, 0, 0)], torch.float32)
        (w_t, w_d), = batch.run()
        assert ops.ps_kind(w_t) is None and w_t._psw is not None and w_t._psw[1] == 13 and w_d._psw[1] == 3
        w_t0, w_d0 = w_t.clone(), w_d.clone()                         # untagged: the library splits them per launch
        x = torch.randn(V, H, H, Cin, device=DEV, generator=g)
        dy = torch.randn(V, H, H, Cout, device=DEV, generator=g)
        M = V * H * H
        y1, _, s1 = ops.conv2d_fwd_with_stats(x, w_t, k, k, 1, pad, H, H, ops.conv_stats(M, Cout, DEV))
        y0, _, s0 = ops.conv2d_fwd_with_stats(x, w_t0, k, k, 1, pad, H, H, ops.conv_stats(M, Cout, DEV))
        d1 = ops.conv2d_dgrad(dy, w_d, k, k, 1, pad, H, H)
        d0 = ops.conv2d_dgrad(dy, w_d0, k, k, 1, pad, H, H)
        sc, sh = torch.rand(Cout, device=DEV, generator=g) + 0.5, torch.randn(Cout, device=DEV, generator=g)
        res = [_res('psw_fwd', y1, y0, 0.0), _res('psw_dgrad', d1, d0, 0.0),
               # the pivot is decoded from the fp16 pieces (2^-22 off the fp32 weights): the raw moments agree to rounding, not bit for bit
               _res('psw_fwd_sums', s1, s0, 5e-6)]
        if k == 1:
            f1 = ops.conv2d_fwd_bn_apply(x, w_t, 1, 1, 1, 0, H, H, sc, sh, res=y0, relu=True)
            f0 = ops.conv2d_fwd_bn_apply(x, w_t0, 1, 1, 1, 0, H, H, sc, sh, res=y0, relu=True)
            res.append(_res('psw_fwd_bn_apply', f1, f0, 0.0))
        # a forward with other terms must not take the copy (an inference forward runs six bf16 terms)
        ops.set_f32_matmul('bf16x6_3')
        y6 = ops.conv2d_fwd(x, w_t, k, k, 1, pad, H, H)
        y6_0 = ops.conv2d_fwd(x, w_t0, k, k, 1, pad, H, H)
        torch.cuda.synchronize()
        res.append(_res('psw_other_terms_fall_back', y6, y6_0, 0.0))
        return res
    finally:
        ops.set_f32_matmul('exact')


def check_ps_weight_pieces(Cin, Cout, k, seed=0):
    """The pieces simclr_presplit_weights_multi stores against torch's own rounding (an oracle independent of the device's split
    instructions): forward copy = fp16 pieces of 2^8 w_t, hi = fp16(x), lo = fp16(x - hi); data-gradient copy = bf16 pieces of w_d.
    The weights span 40 binades, so that lo pieces of every kind occur: normal, subnormal in fp16, zero."""
    ops.set_f32_matmul('f16x3_3')
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        w = torch.randn(k, k, Cin, Cout, device=DEV, generator=g)
        w = w * torch.exp2(torch.randint(-36, 5, w.shape, device=DEV, generator=g).float())
        w.view(-1)[::97] = 0.0
        (w_t, w_d), = ops.WeightPairBatch([(w, 0, 0)], torch.float32).run()
        torch.cuda.synchronize()
        res = []
        for name, wt, kind, scale in (('fwd_f16', w_t, 'f16', 256.0), ('dgrad_bf16', w_d, 'b16', 1.0)):
            ps, terms = wt._psw
            assert terms == (13 if kind == 'f16' else 3)
            _, hi, lo = ps_decode(ps, kind)
            x = (wt.detach() * scale).cpu()
            pt = torch.float16 if kind == 'f16' else torch.bfloat16
            hi_ref = x.to(pt)
            lo_ref = (x - hi_ref.float()).to(pt)
            res += [_res('psw_pieces_hi ' + name, hi, hi_ref.double(), 0.0), _res('psw_pieces_lo ' + name, lo, lo_ref.double(), 0.0)]
            n_sub = int(((lo_ref != 0) & (lo_ref.float().abs() < 2.0 ** -14)).sum()) if kind == 'f16' else 1
            res.append(dict(name='psw_pieces_subnormal_lo_present ' + name, ok=n_sub > 0, err=float(n_sub), tol=0.0))
        return res
    finally:
        ops.set_f32_matmul('exact')


def check_wgrad_workspace_bound(V, H, Cin, Cout, k, matmul='f16x3_3', ps=True, seed=0):
    """simclr_conv2d_wgrad writes its split-K slabs into a caller-provided workspace of simclr_conv2d_wgrad_workspace_bytes: every kernel
    variant the launcher may pick (per-tap, nine-tap with 32-pixel chunks, the 256 x 256 tile) must stay inside it.  The workspace here is
    EXACTLY that size, followed by a poisoned guard region that must come back untouched, and the result must equal the one computed
    with the library's shared (larger) scratch buffer."""
    from simclr_amd._lib import lib
    ops.set_f32_matmul(matmul)
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        pad = (k - 1) // 2
        x = torch.randn(V, H, H, Cin, device=DEV, generator=g)
        dy = torch.randn(V, H, H, Cout, device=DEV, generator=g) * 1e-3
        if ps:
            one, zero = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
            dy, _ = ops.bn_bwd_apply(dy, dy, None, one, zero, zero, one, zero, zero, 0, ps_out=True)
        ref = ops.conv2d_wgrad(x, dy, k, k, 1, pad)
        nbytes = lib().conv2d_wgrad_workspace_bytes(V, H, H, Cin, Cout, k, k, ops.dt(x))
        guard = 1 << 20
        buf = torch.full(((nbytes + 3) // 4 + guard,), 12345.0, device=DEV, dtype=torch.float32)
        out = torch.empty(k * k * Cin, Cout, device=DEV, dtype=torch.float32)
        lib().conv2d_wgrad(ops._p(x), ops._pp(dy), ops._p(out), 0, ops._p(buf), V, H, H, Cin, Cin, H, H, Cout, k, k, 1, pad,
                           ops.dt(x) | (ops.FMT_PS_IN if ops.ps_kind(dy) else 0) | ops._tb(), ops._s())
        torch.cuda.synchronize()
        tag = 'V%d %dx%d %d->%d k%d %s' % (V, H, H, Cin, Cout, k, 'ps' if ps else 'plain')
        tail = buf[(nbytes + 3) // 4:]
        return [_res('wgrad_ws_guard_untouched ' + tag, tail, torch.full_like(tail, 12345.0), 0.0),
                _res('wgrad_ws_same_result ' + tag, out, ref, 0.0)]
    finally:
        ops.set_f32_matmul('exact')


def check_sparse_dgrad(V, H, Cs, Cin, Cmid, mode, seed=0, matmul='f16x3_3'):
    """Stride-2 1x1 projection shortcut + the block's first convolution in the backward pass (fp32 storage): the shortcut's data gradient
    stores only the even (row, column) pixels of dx (accumulate = 3, the rest UNINITIALISED -- poisoned with NaN here) and conv1's
    accumulating data gradient reads earlier data there only (accumulate = 2); against the zero-filling store + full accumulate, bit for
    bit.  mode 0: plain conv1 data gradient; 4: with the fused BatchNorm-backward sums of the previous block's tail."""
    ops.set_f32_matmul(matmul)
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        OH = H // 2
        w_sc = torch.randn(1, 1, Cin, Cs, device=DEV, generator=g) * Cin ** -0.5          # shortcut: Cin -> Cs, stride 2
        w_1 = torch.randn(1, 1, Cin, Cmid, device=DEV, generator=g) * Cin ** -0.5         # conv1: Cin -> Cmid, stride 1
        _, wd_sc = ops.prep_weights_pair(w_sc, torch.float32)
        _, wd_1 = ops.prep_weights_pair(w_1, torch.float32)
        d_sc = torch.randn(V, OH, OH, Cs, device=DEV, generator=g)
        d_1 = torch.randn(V, H, H, Cmid, device=DEV, generator=g)
        mask = (torch.rand(V, H, H, Cin // 4, device=DEV, generator=g) * 16).to(torch.uint8)      # ReLU bits of the previous block's output

        def run(sparse):
            dx = torch.full((V, H, H, Cin), float('nan'), device=DEV)
            ops.conv2d_dgrad(d_sc, wd_sc, 1, 1, 2, 0, H, H, out=dx, sparse=sparse)
            if mode == 0:
                ops.conv2d_dgrad(d_1, wd_1, 1, 1, 1, 0, H, H, out=dx, accumulate=True)
                return dx, None
            out, part = ops.conv2d_dgrad_bn(d_1, wd_1, 1, 1, 0, H, H, dict(mask=mask.view(V * H * H, -1), mode=4), out=dx, accumulate=True)
            return out, ops.bn_reduce_slots(part)

        a, pa = run(False)
        b, pb = run(True)
        torch.cuda.synchronize()
        tag = 'V%d %d %d/%d/%d mode%d' % (V, H, Cs, Cin, Cmid, mode)
        res = [_res('sparse_dgrad_dx ' + tag, b, a, 0.0)]
        if pa is not None:
            res.append(_res('sparse_dgrad_sums ' + tag, pb, pa, 0.0))
        return res
    finally:
        ops.set_f32_matmul('exact')


def ps_encode(t):
    """float32 [..., C] (C % 32 == 0) -> the pre-split block format with bf16 pieces (the inverse of ps_decode; hi = bf16(x) to nearest
    even, lo = bf16(x - hi)), tagged `_ps` like the tensors simclr_bn_bwd_apply writes."""
    C = t.shape[-1]
    x = t.detach().float().cpu().reshape(-1, C // 32, 32)
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    ch = torch.tensor([[4 * g + i if i < 4 else 16 + 4 * g + i - 4 for i in range(8)] for g in range(4)]).reshape(-1)
    blk = torch.cat([hi[:, :, ch], lo[:, :, ch]], dim=2).contiguous()           # [rows, blocks, 64 pieces] = 128 bytes
    out = blk.view(torch.int16).view(torch.float32).reshape(t.shape).contiguous().to(t.device)
    out._ps = 'b16'
    return out


def check_ps_backward(V, H, Cin, Cout, k, stride, seed=0, matmul='bf16x3'):
    """The pre-split gradient path of the fp32 parity mode (round 6): simclr_bn_bwd_apply writes dx as (hi, lo) bf16 pieces per 128-byte
    block, and the data-gradient / weight-gradient GEMMs of the convolution in front of that BatchNorm read the pieces without splitting
    anything in their k-loops.  Checked: the stored pieces are exactly bf16(dx) and bf16(dx - hi) of the plain kernel's dx; the data
    gradient (plain, and with the fused BatchNorm-backward reduce) is BITWISE the in-register-split result; the weight gradient (different
    pixel order inside a k-step) against float64 of the three-term operands."""
    ops.set_f32_matmul(matmul)
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        pad = (k - 1) // 2
        OH = (H + (k - 1) - k) // stride + 1
        OW = (H + (k - 1) - k) // stride + 1
        rnd = lambda shape, sc=1.0: torch.randn(shape, device=DEV, generator=g) * sc
        x = rnd((V, H, H, Cin))
        w = rnd((k, k, Cin, Cout), (k * k * Cin) ** -0.5)
        w_d = ops.prep_weights(w, 1, torch.float32)
        # a BatchNorm backward apply over the conv OUTPUT geometry [V, OH, OW, Cout] produces the gradient the conv's backward consumes
        dy_in = rnd((V, OH, OW, Cout), 1e-3)
        xc = rnd((V, OH, OW, Cout), 1.5) + 0.3
        scale = torch.rand(Cout, device=DEV, generator=g) + 0.2
        shift = 0.3 * rnd((Cout,))
        mean = 0.2 * rnd((Cout,))
        rstd = 0.5 + torch.rand(Cout, device=DEV, generator=g)
        c1 = 1e-4 * rnd((Cout,))
        c2 = 1e-4 * rnd((Cout,))
        dx_plain, _ = ops.bn_bwd_apply(dy_in, xc, None, scale, shift, mean, rstd, c1, c2, 2)
        dx_ps, _ = ops.bn_bwd_apply(dy_in, xc, None, scale, shift, mean, rstd, c1, c2, 2, ps_out=True)
        assert ops.ps_kind(dx_ps) == 'b16'
        tag = 'V%d %dx%d %d->%d k%d s%d' % (V, H, H, Cin, Cout, k, stride)
        val, hi, lo = ps_decode(dx_ps)
        ref = dx_plain.double().cpu()
        hi_ref = dx_plain.bfloat16().double().cpu()
        lo_ref = (dx_plain - dx_plain.bfloat16().float()).bfloat16().double().cpu()
        res = [_res('ps_pieces_hi ' + tag, hi, hi_ref, 0.0), _res('ps_pieces_lo ' + tag, lo, lo_ref, 0.0),
               _res('ps_value ' + tag, val, ref, 2.0 ** -16)]
        # data gradient: bitwise the in-register split -- except on the 3x3 stride-1 layers, where the pre-split launch takes the halo-window
        # path (same products, summed chunk-major instead of tap-major) and the in-register split gathers: three-term rounding level there
        win = k == 3 and stride == 1 and H <= 62
        bit = 4e-5 if win else 0.0
        d_plain = ops.conv2d_dgrad(dx_plain, w_d, k, k, stride, pad, H, H)
        d_ps = ops.conv2d_dgrad(dx_ps, w_d, k, k, stride, pad, H, H)
        res.append(_res('ps_dgrad_bitwise ' + tag, d_ps, d_plain, bit))
        if stride == 1:
            bn_x = rnd((V, H, H, Cin), 1.5) + 0.3
            bn = dict(x=bn_x, mask=None, scale=torch.rand(Cin, device=DEV, generator=g) - 0.4, shift=0.3 * rnd((Cin,)),
                      mean=0.2 * rnd((Cin,)), rstd=0.5 + torch.rand(Cin, device=DEV, generator=g), mode=2)
            m_plain, p_plain = ops.conv2d_dgrad_bn(dx_plain, w_d, k, k, pad, H, H, bn)
            m_ps, p_ps = ops.conv2d_dgrad_bn(dx_ps, w_d, k, k, pad, H, H, bn)
            res.append(_res('ps_dgrad_bn_bitwise ' + tag, m_ps, m_plain, bit))
            res.append(_res('ps_dgrad_bn_sums_bitwise ' + tag, ops.bn_reduce_slots(p_ps), ops.bn_reduce_slots(p_plain), 25 * bit))
        # weight gradient: float64 of the exact operands (the three-term gate of check_conv), and close to the plain kernel
        dw_plain = ops.conv2d_wgrad(x, dx_plain, k, k, stride, pad)
        dw_ps = ops.conv2d_wgrad(x, dx_ps, k, k, stride, pad)
        torch.cuda.synchronize()
        xr = x.double().permute(0, 3, 1, 2)
        wr = w.double().requires_grad_(True)
        pe = (k - 1) - pad
        yr = F.conv2d(F.pad(xr, (pad, pe, pad, pe)), wr.permute(3, 2, 0, 1), stride=stride)
        yr.backward(dx_plain.double().permute(0, 3, 1, 2))
        res.append(_res('ps_wgrad_vs_f64 ' + tag, dw_ps.view(k, k, Cin, Cout), wr.grad, 4e-5))
        res.append(_res('ps_wgrad_vs_plain ' + tag, dw_ps, dw_plain, 4e-5))
        return res
    finally:
        ops.set_f32_matmul('exact')


def check_conv_bench_path(V, H, Cin, Cout, k, stride, dtype, seed=0, nsample=4096, bn_case=None, bwd_tol_scale=1.0,
                          rounded_stats=False):
    """Forward / dgrad / wgrad (and, for stride 1, the fused dgrad + BN-backward reduce) at BASELINE cfg2 layer
    shapes with enough rows that every persistent workgroup walks several tiles -- the regime bench.py runs.
    References: (a) plain-torch float64 on the device over the FULL tensors (incl. dW and the BN sums),
    (b) float64 on the CPU over `nsample` random output rows, computed independently by gathering patches.
    bn_case: None | (mask_mode, accumulate) for simclr_conv2d_dgrad_bn."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = (k - 1) // 2
    OH = (H + (k - 1) - k) // stride + 1
    OW = OH
    W = H

    def rnd(shape, scale=1.0):
        return (torch.randn(shape, device=DEV, generator=g) * scale).to(dtype)

    x = rnd((V, H, W, Cin))
    w = rnd((k, k, Cin, Cout), (k * k * Cin) ** -0.5)
    dy = rnd((V, OH, OW, Cout))
    w32 = w.float()
    w_t = ops.prep_weights(w32, 0, dtype)
    w_d = ops.prep_weights(w32, 1, dtype)
    stats = ops.conv_stats(V * OH * OW, Cout, DEV)          # one slot per workgroup: the deterministic path of the step
    y, y_guard = _guarded((V, OH, OW, Cout), dtype)
    dx, dx_guard = _guarded((V, H, W, Cin), dtype)
    dw, dw_guard = _guarded((k * k * Cin, Cout), torch.float32)
    ops.conv2d_fwd(x, w_t, k, k, stride, pad, OH, OW, stats=stats, out=y)
    sums = ops.bn_reduce_slots(stats)
    stats_b = ops.conv_stats(V * OH * OW, Cout, DEV)        # run-to-run determinism of the statistics (bitwise)
    ops.conv2d_fwd(x, w_t, k, k, stride, pad, OH, OW, stats=stats_b, out=torch.empty_like(y))
    sums_b = ops.bn_reduce_slots(stats_b)
    ops.conv2d_dgrad(dy, w_d, k, k, stride, pad, H, W, out=dx)
    ops.conv2d_wgrad(x, dy, k, k, stride, pad, out=dw)
    dm = part = None
    if bn_case is not None:
        mode, acc = bn_case
        epc = 8 if dtype == torch.bfloat16 else 4
        bn_x = rnd((V, H, W, Cin), 1.5) + 0.3
        scale = torch.rand(Cin, device=DEV, generator=g) - 0.4
        shift = 0.3 * torch.randn(Cin, device=DEV, generator=g)
        mean = 0.2 * torch.randn(Cin, device=DEV, generator=g)
        rstd = 0.5 + torch.rand(Cin, device=DEV, generator=g)
        prev = rnd((V, H, W, Cin)) if acc else None
        mask_t = rnd((V, H, W, Cin)) if mode in (1, 3) else None
        mask_arg = mask_t
        if mode == 3:
            mb = (mask_t > 0).reshape(-1, Cin // epc, epc).to(torch.int32)
            mask_arg = (mb << torch.arange(epc, dtype=torch.int32, device=DEV)).sum(-1).to(torch.uint8)
        dm, dm_guard = _guarded((V, H, W, Cin), dtype, fill=prev)
        bn = dict(x=bn_x, mask=mask_arg, scale=scale, shift=shift, mean=mean, rstd=rstd, mode=mode)
        _, part = ops.conv2d_dgrad_bn(dy, w_d, k, k, pad, H, W, bn, out=dm, accumulate=acc)
        bsum = ops.bn_reduce_slots(part)
    torch.cuda.synchronize()

    tag = 'V%d %dx%d %d->%d k%d s%d %s' % (V, H, W, Cin, Cout, k, stride, str(dtype).split('.')[-1])
    t = _tol(dtype)
    res = []
    oob = y_guard() + dx_guard() + dw_guard() + (dm_guard() if dm is not None else 0)
    res.append(dict(name='bigconv_guard_bands ' + tag, err=float(oob), tol=0.0, scale=0.0, ok=oob == 0, nbad=oob, numel=4))
    same = bool(torch.equal(sums, sums_b))
    res.append(dict(name='bigconv_stats_bitwise_repeatable ' + tag, err=0.0 if same else 1.0, tol=0.0, scale=0.0, ok=same, nbad=0, numel=1))

    # ---- (a) full tensors vs float64 torch on the device
    vchunk = max(1, min(V, int(3e8 // (H * W * max(Cin, Cout) * max(1, k * k // 3)))))
    e_y = e_dx = e_dm = 0.0
    m_y = m_dx = m_dm = 0.0
    s1 = torch.zeros(Cout, device=DEV, dtype=torch.float64)
    s2 = torch.zeros_like(s1)
    b1 = torch.zeros(Cin, device=DEV, dtype=torch.float64)
    b2 = torch.zeros_like(b1)
    l1 = torch.zeros_like(b1)
    xh_max = 0.0
    dw64 = None
    for v0, v1, y64, dx64, dw64 in _ref64_conv(x, w, dy, k, stride, pad, OH, OW, vchunk):
        e_y = max(e_y, float((y[v0:v1].double() - y64).abs().max())); m_y = max(m_y, float(y64.abs().max()))
        e_dx = max(e_dx, float((dx[v0:v1].double() - dx64).abs().max())); m_dx = max(m_dx, float(dx64.abs().max()))
        s1 += y64.sum((0, 1, 2)); s2 += (y64 * y64).sum((0, 1, 2))
        if dm is not None:
            da = dx64 + (prev[v0:v1].double() if acc else 0.0)
            if mode in (1, 3):
                mk = mask_t[v0:v1] > 0
            else:   # the kernel's fmaf(x, scale, shift) > 0: the exact sign, which float64 arithmetic reproduces
                mk = (bn_x[v0:v1].double() * scale.double() + shift.double()) > 0
            dmr = torch.where(mk, da, torch.zeros((), device=DEV, dtype=torch.float64))
            xh = (bn_x[v0:v1].double() - mean.double()) * rstd.double()
            xh_max = max(xh_max, float(xh.abs().max()))
            b1 += dmr.sum((0, 1, 2)); b2 += (dmr * xh).sum((0, 1, 2)); l1 += (dmr * xh).abs().sum((0, 1, 2)) + dmr.abs().sum((0, 1, 2))
            e_dm = max(e_dm, float((dm[v0:v1].double() - dmr).abs().max())); m_dm = max(m_dm, float(dmr.abs().max()))
        del y64, dx64

    def ent(name, err, scale, rtol, atol=0.0):
        tol = rtol * scale + atol
        return dict(name=name + ' ' + tag, err=float(err), tol=float(tol), scale=float(scale), ok=bool(err <= tol), nbad=0, numel=1)

    res.append(ent('bigconv_fwd_full', e_y, m_y, t))
    tb = t * bwd_tol_scale          # three-term split-bf16 backward arithmetic (simclr_set_f32_matmul): 2 x the fp32 gate
    res.append(ent('bigconv_dgrad_full', e_dx, m_dx, tb))
    sum_tol = 1e-5 if dtype == torch.float32 else 1e-4     # relative to the L1 mass of the summed terms
    # rounded_stats (256-wide bf16 tiles): the sums are taken over the bf16-ROUNDED outputs -- against the unrounded float64
    # sums a random walk of M half-ulp steps (the allowance the fused BN-backward sums below get), which dominates for small M
    rw = 6.0 * (V * OH * OW) ** 0.5 * 2.0 ** -9 * m_y if (rounded_stats and dtype == torch.bfloat16) else 0.0
    res.append(ent('bigconv_stats_sum', float((sums[0] - s1).abs().max()), float(s2.max()) ** 0.5 * (V * OH * OW) ** 0.5, sum_tol, rw))
    res.append(ent('bigconv_stats_sq', float((sums[1] - s2).abs().max()), float(s2.max()), sum_tol, 2.0 * rw * m_y))
    res.append(ent('bigconv_wgrad_full', float((dw.view(k, k, Cin, Cout).double() - dw64).abs().max()), float(dw64.abs().max()),
                   (2e-5 if dtype == torch.float32 else 1e-4) * bwd_tol_scale))
    if dm is not None:
        btag = ' mode%d acc%d' % (mode, acc)
        res.append(ent('bigconv_dgrad_bn_dm' + btag, e_dm, m_dm, tb * (2 if acc else 1)))
        # bf16: the kernel sums the ROUNDED dm it stores (what the BatchNorm backward apply will read); against the unrounded
        # float64 sums that is a random walk of M steps of half-ulp size, which dominates the L1-relative term for small M
        walk = 0.0 if dtype == torch.float32 else 6.0 * (V * H * W) ** 0.5 * 2.0 ** -9 * m_dm
        res.append(ent('bigconv_dgrad_bn_sum' + btag, float((bsum[0] - b1).abs().max()), float(l1.max()), sum_tol, walk))
        res.append(ent('bigconv_dgrad_bn_sumxhat' + btag, float((bsum[1] - b2).abs().max()), float(l1.max()), sum_tol,
                       walk * max(1.0, xh_max)))

    # ---- (b) sampled rows vs float64 on the CPU (independent gather formulation)
    gs = torch.Generator().manual_seed(seed + 99)
    wc = w.double().cpu()
    # forward: output pixel (v, oy, ox) <- patch of x
    mo = torch.randint(0, V * OH * OW, (nsample,), generator=gs)
    vv, rem = mo // (OH * OW), mo % (OH * OW)
    oy, ox = rem // OW, rem % OW
    patch = torch.zeros(nsample, k, k, Cin, dtype=torch.float64)
    xc = None
    for ty in range(k):
        for tx in range(k):
            iy, ix = oy * stride - pad + ty, ox * stride - pad + tx
            ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
            idx = ((vv * H + iy.clamp(0, H - 1)) * W + ix.clamp(0, W - 1)).to(DEV)
            rows = x.view(-1, Cin)[idx].double().cpu()
            patch[:, ty, tx] = torch.where(ok[:, None], rows, torch.zeros((), dtype=torch.float64))
    y_cpu = torch.einsum('nabc,abcd->nd', patch, wc)
    y_got = y.view(-1, Cout)[mo.to(DEV)].double().cpu()
    res.append(_res('bigconv_fwd_cpu_rows ' + tag, y_got, y_cpu, t))
    # dgrad: input pixel (v, iy, ix) <- every (tap, output pixel) that read it
    mi = torch.randint(0, V * H * W, (nsample,), generator=gs)
    vv, rem = mi // (H * W), mi % (H * W)
    iy, ix = rem // W, rem % W
    dx_cpu = torch.zeros(nsample, Cin, dtype=torch.float64)
    for ty in range(k):
        for tx in range(k):
            ny, nx = iy + pad - ty, ix + pad - tx
            ok = (ny >= 0) & (nx >= 0) & (ny % stride == 0) & (nx % stride == 0)
            qy, qx = ny // stride, nx // stride
            ok = ok & (qy < OH) & (qx < OW)
            idx = ((vv * OH + qy.clamp(0, OH - 1)) * OW + qx.clamp(0, OW - 1)).to(DEV)
            rows = dy.view(-1, Cout)[idx].double().cpu()
            rows = torch.where(ok[:, None], rows, torch.zeros((), dtype=torch.float64))
            dx_cpu += rows @ wc[ty, tx].t()
    dx_got = dx.view(-1, Cin)[mi.to(DEV)].double().cpu()
    res.append(_res('bigconv_dgrad_cpu_rows ' + tag, dx_got, dx_cpu, tb))
    return res


_STEP_ORACLE_CACHE = {}


def structured_images(batch, size, views, gen):
    """Synthetic images that DIFFER from each other the way photographs do: a random base colour per image and view plus a
    few random low-frequency waves per channel.  i.i.d. uniform noise (the benchmark input) makes every image
    statistically identical, so after the first pooling the features are nearly the same for the whole batch and
    every BatchNorm over the batch amplifies pure rounding noise -- fine for timing, useless for judging precision."""
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing='ij')
    C = 3 * views
    base = 0.2 + 0.6 * torch.rand(batch, 1, 1, C, generator=gen)
    img = base.expand(batch, size, size, C).clone()
    for _ in range(4):
        fy = 6.0 * torch.rand(batch, 1, 1, C, generator=gen)
        fx = 6.0 * torch.rand(batch, 1, 1, C, generator=gen)
        ph = 6.2832 * torch.rand(batch, 1, 1, C, generator=gen)
        amp = 0.25 * torch.rand(batch, 1, 1, C, generator=gen)
        img += amp * torch.sin(6.2832 * (fy * yy[None, :, :, None] + fx * xx[None, :, :, None]) + ph)
    img += 0.05 * torch.rand(batch, size, size, C, generator=gen)
    return img.clamp_(0.0, 1.0)


def correlated_views(batch, image_size, gen):
    """[b, S, S, 6]: two correlated views per image (shift / flip / brightness / noise of one structured image), so the
    contrastive task is learnable -- a stand-in for tf2/data_util.py's two-view augmentation with fixed draws."""
    base = structured_images(batch, image_size + 4, 1, gen)                       # [b, S+4, S+4, 3]
    v1 = base[:, 2:2 + image_size, 2:2 + image_size]
    dx = int(torch.randint(0, 5, (1,), generator=gen)); dy = int(torch.randint(0, 5, (1,), generator=gen))
    v2 = base[:, dy:dy + image_size, dx:dx + image_size].flip(2)
    v2 = v2 * (0.8 + 0.4 * torch.rand(batch, 1, 1, 1, generator=gen))
    two = torch.cat([v1 + 0.03 * torch.randn(v1.shape, generator=gen), v2 + 0.03 * torch.randn(v2.shape, generator=gen)], 3)
    return two.clamp_(0.0, 1.0).contiguous()


def check_train_step_fixed(depth=50, image_size=224, batch=32, compute_dtype='f32', num_classes=1000, seed=0,
                           weight_decay=1e-6, lr=0.1, head_dtype='same', inputs='structured', randomize_bn=False,
                           gates=None, pretrain_steps=0, pretrain_lr=0.3, pretrain_pool=8, f32_matmul='exact'):
    """pretrain_steps > 0: the step under test starts from a TRAINED point instead of the
    initialisation -- the network is first trained on the device for `pretrain_steps` steps (fp32 parity mode, correlated
    two-view batches) until the contrastive task is solved (features differ from image to image, gamma != 0 on the block
    tails), its weights and BatchNorm statistics are exported to the float64 oracle, and one step on a held-out
    correlated batch is compared.  `gates` overrides thresholds by name.  (randomize_bn=True -- random gamma / beta on
    every BatchNorm of an UNTRAINED network -- is kept as a diagnostic only: an untrained deep network maps every image to
    nearly the same feature, and with gamma != 0 on all 16 tails the per-image signal under the common mode is so small
    that even the fp32 mode misses its gates 6-20x)

    One full pretraining step at a realistic batch (BatchNorm well conditioned) with the reference
    initialisation, against the float64 oracle, gated by FIXED thresholds (no calibration):
      f32 : BASELINE.json north_star -- loss <= 1e-3 rel, normalised embeddings <= 1e-5 abs; plus gradient
            1-cos <= 1e-6, every gradient tensor within 1e-3 of the GLOBAL gradient norm, new weights <= 1e-5 rel.
      bf16: loss <= 1e-2 rel, gradient 1-cos <= 2e-2 (measured 1.0e-2 on image-like inputs, 1.9e-2 on i.i.d. noise: the
            ill-conditioned start of training -- zero-initialised residual branches, BatchNorm over 64 nearly identical
            feature rows -- amplifies the 2^-9 storage rounding ~50x; fp32 heads / fp32 pooled features do not change it,
            profiles/r02_bf16_parity.json), embeddings reported and bounded at 5e-2 abs.
    The oracle step is computed once per configuration and shared by the f32 and bf16 cases."""
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, train_step
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step

    key = (depth, image_size, batch, num_classes, seed, weight_decay, lr, inputs, randomize_bn, pretrain_steps, pretrain_lr)
    if key not in _STEP_ORACLE_CACHE:
        cfg = Config(resnet_depth=depth, image_size=image_size, num_classes=num_classes, weight_decay=weight_decay)
        params, state = init_model(cfg, seed=seed, randomize_bn=randomize_bn)
        momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        g = torch.Generator().manual_seed(seed + 1)
        images = (torch.rand(batch, image_size, image_size, 6, generator=g) if inputs == 'iid'
                  else correlated_views(batch, image_size, g) if pretrain_steps > 0
                  else structured_images(batch, image_size, 2, g))
        labels = torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float()
        if pretrain_steps > 0:
            pool = [(correlated_views(batch, image_size, g).to(DEV),
                     {'labels': torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float().to(DEV)})
                    for _ in range(pretrain_pool)]
            FLAGS.reset()
            FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype='f32', use_blur=False,
                         weight_decay=weight_decay, train_batch_size=batch)
            RT.reset()
            RT.device = torch.device(DEV)
            pm = model_lib.Model(num_classes)
            with torch.no_grad():
                pm(torch.zeros(2, image_size, image_size, 6, device=DEV), training=True)
            allv = dict(params); allv.update(state)
            for v in pm.variables:
                v.value.copy_(allv[v.name].to(DEV))
            RT.weights_version += 1
            pstep = make_single_step(pm, model_lib.build_optimizer(pretrain_lr), None)
            for i in range(pretrain_steps):
                pout = pstep(*pool[i % pretrain_pool])
            torch.cuda.synchronize()
            print('   pretrained %d steps: contrast loss %.4f, contrast accuracy %.3f' % (
                pretrain_steps, float(pout['con_loss'].value.reshape(-1)[0]), float(pout['logits_con'].contrast_acc.reshape(-1)[0])))
            trained = {v.name: v.value.detach().float().cpu().clone() for v in pm.variables}
            params = OrderedDict((k, trained[k].reshape(v.shape)) for k, v in params.items())
            state = OrderedDict((k, trained[k].reshape(v.shape)) for k, v in state.items())
            del pm, pstep, pool
            torch.cuda.empty_cache()
        p64 = OrderedDict((k, v.double()) for k, v in params.items())
        s64 = OrderedDict((k, v.double()) for k, v in state.items())
        m64 = OrderedDict((k, v.double()) for k, v in momenta.items())
        np64, ns64, nm64, t64 = train_step(cfg, p64, s64, m64, images.double(), labels.double(), lr)
        _STEP_ORACLE_CACHE[key] = (params, state, images, labels, np64, ns64, t64)
    params, state, images, labels, np64, ns64, t64 = _STEP_ORACLE_CACHE[key]

    FLAGS.reset()
    FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype=compute_dtype, use_blur=False,
                 weight_decay=weight_decay, train_batch_size=batch, head_dtype=head_dtype, f32_matmul=f32_matmul)
    RT.reset()
    RT.device = torch.device(DEV)
    model = model_lib.Model(num_classes)
    with torch.no_grad():
        model(torch.zeros(2, image_size, image_size, 6, device=DEV), training=True)
    allv = dict(params); allv.update(state)
    for v in model.variables:
        v.value.copy_(allv[v.name].to(DEV))
    RT.weights_version += 1
    optimizer = model_lib.build_optimizer(lr)
    step_fn = make_single_step(model, optimizer, None)
    out = step_fn(images.to(DEV), {'labels': labels.to(DEV)})
    torch.cuda.synchronize()
    emu = compute_dtype == 'bf16'
    tag = 'R%d %dpx b%d %s%s %s%s%s fixed' % (depth, image_size, batch, compute_dtype + ('' if f32_matmul == 'exact' else '/' + f32_matmul), '' if head_dtype == 'same' else '+head_' + head_dtype,
                                              inputs, ' randbn' if randomize_bn else '', ' trained%d' % pretrain_steps if pretrain_steps else '')
    res = []

    def gate(name, err, tol, **kw):
        if gates and name in gates:
            tol = gates[name]
        d = dict(name='%s %s' % (name, tag), err=float(err), tol=float(tol), scale=1.0, ok=bool(err <= tol), nbad=0, numel=1)
        d.update(kw)
        res.append(d)

    def rel(a, b):
        return float((a.double().cpu() - b.double()).abs().max()) / (float(b.double().abs().max()) + 1e-30)

    gate('fixed_con_loss_rel', rel(out['con_loss'].value.reshape(-1)[0], t64['con_loss'].detach()), 1e-2 if emu else 1e-3,
         value=float(out['con_loss'].value.reshape(-1)[0]), ref=float(t64['con_loss']))
    gate('fixed_sup_loss_rel', rel(out['sup_loss'].value.reshape(-1)[0], t64['sup_loss'].detach()), 1e-2 if emu else 1e-3)
    z_err = float((out['con_loss'].normalized.double().cpu() - t64['z'].detach()).abs().max())
    gate('fixed_embeddings_abs', z_err, 5e-2 if emu else 1e-5)
    keys = list(params.keys())
    byname = {v.name: v for v in model._flat_order}
    g64 = torch.cat([(t64['grads'][k] if t64['grads'][k] is not None else torch.zeros_like(np64[k])).double().reshape(-1) for k in keys])
    gm = torch.cat([byname[k].grad.double().reshape(-1).cpu() for k in keys])
    gate('fixed_grad_1-cos', 1.0 - float((gm * g64).sum() / gm.norm() / g64.norm()), 2e-2 if emu else 1e-6)
    gate('fixed_grad_relnorm', float((gm - g64).norm() / g64.norm()), 2e-1 if emu else 2e-3)
    gn = float(g64.norm())
    worst, wn = 0.0, ''
    for k in keys:
        ref = t64['grads'][k]
        if ref is None:
            continue
        e = float((byname[k].grad.double().cpu() - ref).norm()) / gn
        if e > worst:
            worst, wn = e, k
    gate('fixed_grad_tensor_vs_global_norm', worst, 1.5e-1 if emu else 1e-3, worst=wn)
    # where the gradient error sits: the tensors with the largest share of |g - g_ref|^2 (diagnostic print)
    contrib = []
    for k in keys:
        ref = t64['grads'][k]
        if ref is None:
            continue
        d2 = float((byname[k].grad.double().cpu() - ref).norm()) ** 2
        contrib.append((d2, k, float(ref.norm()), float(byname[k].grad.double().norm())))
    contrib.sort(reverse=True)
    tot = sum(c[0] for c in contrib) + 1e-300
    for d2, k, rn, mn in contrib[:8]:
        print('   grad-error share %5.1f%%  %-72s |ref|=%.3e |mine|=%.3e rel=%.3e' % (100 * d2 / tot, k, rn, mn, d2 ** 0.5 / (rn + 1e-30)))
    # the LARS update (tf2/lars_optimizer.py:83-137) normalises every tensor's step by its own gradient norm, so the
    # update error is gated globally (|dw - dw_ref| / |dw_ref| over all weights), not per tensor relative to the weight
    num = den = 0.0
    for k in keys:
        old = params[k].double()
        dw_ref = np64[k] - old
        dw = byname[k].value.double().cpu() - old
        num += float((dw - dw_ref).norm()) ** 2
        den += float(dw_ref.norm()) ** 2
    gate('fixed_update_relnorm', (num / (den + 1e-300)) ** 0.5, 0.5 if emu else 1e-2)
    bm = max(rel(v.value, ns64[v.name]) for v in model.variables if v.name in ns64)
    gate('fixed_bn_moving_worst_rel', bm, 1e-2 if emu else 1e-5)
    return res


def check_step_at_baseline_size(depth=50, image_size=224, batch=128, f32_matmul='f16x3_3', sk_ratio=0.0, width_multiplier=1,
                                num_classes=1000, seed=0, forward_only=False, lr=0.1, weight_decay=1e-6):
    """parity at BASELINE.json sizes -- the persistent-grid / split-tail / tile decisions of the BENCH shapes end to end in
    the fast parity mode.  One product step (or, forward_only, one training forward) against the torch-CPU float32 oracle (float64 of these
    sizes does not fit the suite's budget; the fp32 oracle's own distance from float64 is ~1e-6 on these quantities, tests at batch 32):
    loss <= 1e-3 relative and l2-normalised embeddings <= 1e-5 absolute (north_star's tolerances), gradient 1 - cos <= 1e-5 and relative
    L2 <= 5e-3 over all trainable tensors.  Reference initialisation, image-like inputs."""
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, single_step_losses, train_step
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step
    cfg = Config(resnet_depth=depth, image_size=image_size, num_classes=num_classes, weight_decay=weight_decay, sk_ratio=sk_ratio,
                 width_multiplier=width_multiplier)
    params, state = init_model(cfg, seed=seed, randomize_bn=False)
    g = torch.Generator().manual_seed(seed + 1)
    images = structured_images(batch, image_size, 2, g)
    labels = torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float()
    if forward_only:
        with torch.no_grad():
            t32 = single_step_losses(cfg, params, state, images, labels)
    else:
        momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        _, _, _, t32 = train_step(cfg, params, state, momenta, images, labels, lr)
    FLAGS.reset()
    FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype='f32', use_blur=False, weight_decay=weight_decay,
                 train_batch_size=batch, f32_matmul=f32_matmul, sk_ratio=sk_ratio, width_multiplier=width_multiplier)
    RT.reset()
    RT.device = torch.device(DEV)
    model = model_lib.Model(num_classes)
    with torch.no_grad():
        model(torch.zeros(2, image_size, image_size, 6, device=DEV), training=True)
    allv = dict(params); allv.update(state)
    for v in model.variables:
        v.value.copy_(allv[v.name].to(DEV))
    RT.weights_version += 1
    tag = 'R%d%s%s %dpx b%d f32/%s%s' % (depth, '' if width_multiplier == 1 else ' %dx' % width_multiplier, ' SK' if sk_ratio > 0 else '',
                                         image_size, batch, f32_matmul, ' forward' if forward_only else '')
    res = []

    def gate(name, err, tol):
        res.append(dict(name='%s %s' % (name, tag), err=float(err), tol=float(tol), scale=1.0, ok=bool(err <= tol), nbad=0, numel=1))

    if forward_only:
        from simclr_amd import objective as obj_lib
        ops.begin_step(torch.device(DEV))
        proj, sup = model(images.to(DEV), training=True)
        con, _, _ = obj_lib.add_contrastive_loss(proj, hidden_norm=True, temperature=FLAGS.temperature, strategy=None)
        torch.cuda.synchronize()
        gate('baseline_con_loss_rel', abs(float(con.value.reshape(-1)[0]) - float(t32['con_loss'])) / abs(float(t32['con_loss'])), 1e-3)
        gate('baseline_embeddings_abs', float((con.normalized.double().cpu() - t32['z'].double()).abs().max()), 1e-5)
        gate('baseline_sup_logits_rel', float((sup.dense().double().cpu() - t32['sup_logits'].double()).abs().max()) /
             (float(t32['sup_logits'].double().abs().max()) + 1e-30), 1e-3)
        ops.end_step()
    else:
        step_fn = make_single_step(model, model_lib.build_optimizer(lr), None)
        out = step_fn(images.to(DEV), {'labels': labels.to(DEV)})
        torch.cuda.synchronize()
        gate('baseline_con_loss_rel', abs(float(out['con_loss'].value.reshape(-1)[0]) - float(t32['con_loss'])) / abs(float(t32['con_loss'])), 1e-3)
        gate('baseline_sup_loss_rel', abs(float(out['sup_loss'].value.reshape(-1)[0]) - float(t32['sup_loss'])) / abs(float(t32['sup_loss'])), 1e-3)
        gate('baseline_embeddings_abs', float((out['con_loss'].normalized.double().cpu() - t32['z'].detach().double()).abs().max()), 1e-5)
        keys = list(params.keys())
        byname = {v.name: v for v in model._flat_order}
        g32 = torch.cat([(t32['grads'][k] if t32['grads'][k] is not None else torch.zeros_like(params[k])).double().reshape(-1) for k in keys])
        gm = torch.cat([byname[k].grad.double().reshape(-1).cpu() for k in keys])
        gate('baseline_grad_1-cos', 1.0 - float((gm * g32).sum() / gm.norm() / g32.norm()), 1e-5)
        gate('baseline_grad_relnorm', float((gm - g32).norm() / g32.norm()), 5e-3)
    del model
    FLAGS.reset()
    RT.reset()
    torch.cuda.empty_cache()
    return res


def check_step_determinism(depth=18, image_size=32, batch=16, compute_dtype='bf16', steps=2, num_classes=10, seed=0,
                           env_second=None, env_both=None):
    """Two fresh models, same weights, same batches: every weight, BN moving statistic and LARS momentum must be
    BIT-IDENTICAL after `steps` steps (the reference's step is deterministic on TPU, tf2/resnet.py:54-60)."""
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step
    g = torch.Generator().manual_seed(seed)
    feats = [torch.rand(batch, image_size, image_size, 6, generator=g).to(DEV) for _ in range(steps)]
    labs = [{'labels': torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float().to(DEV)}
            for _ in range(steps)]
    snaps = []
    import os
    for run in range(2):
        # env_second: environment switches applied to the SECOND run only -- an optimisation that claims to be bitwise
        # neutral (e.g. SIMCLR_CONV3_FUSED=0 vs the default fused conv3 + bn3 forward) must leave every weight identical
        saved_env = {}
        envs = dict(env_both or {})
        if run == 1 and env_second:
            envs.update(env_second)
        if envs:
            for k_, v_ in envs.items():
                saved_env[k_] = os.environ.get(k_)
                os.environ[k_] = v_
        FLAGS.reset()
        FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype=compute_dtype, use_blur=False, train_batch_size=batch)
        RT.reset()
        RT.device = torch.device(DEV)
        RT.seed = 1234                       # same initial weights in both runs
        model = model_lib.Model(num_classes)
        opt = model_lib.build_optimizer(0.1)
        step = make_single_step(model, opt, None)
        for i in range(steps):
            step(feats[i], labs[i])
        torch.cuda.synchronize()
        snap = {v.name: v.value.clone() for v in model.variables}
        snap.update({'momentum/' + v.name: opt.get_slot(v, 'Momentum').clone() for v in model._flat_order})
        snaps.append(snap)
        for k_, v_ in saved_env.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
    diff = [k for k in snaps[0] if not torch.equal(snaps[0][k], snaps[1][k])]
    worst = max([float((snaps[0][k] - snaps[1][k]).abs().max()) for k in diff] + [0.0])
    tag = 'R%d %dpx b%d %s %d steps%s' % (depth, image_size, batch, compute_dtype, steps, ' vs %s' % env_second if env_second else '')
    return [dict(name='step_bitwise_deterministic ' + tag, err=float(len(diff)), tol=0.0, scale=worst, ok=not diff, nbad=len(diff),
                 numel=len(snaps[0]), first=diff[:3])]


def check_bf16_trajectory(depth=18, image_size=32, batch=256, steps=100, num_classes=10, seed=0, pool=8, lr=0.3,
                          after=20, loss_rel_tol=1e-2, acc_tol=2e-2, window=10, yardstick=True):
    """bf16 speed mode vs fp32 parity mode over a TRAINING RUN, not one step from initialisation:
    BASELINE configs[0]'s shape (ResNet-18, 32 px, batch 256), the same initial weights and the same `steps` batches in both
    modes, on the device.  Two correlated views per image (shift / flip / brightness / noise of one structured image), so
    the contrastive task is learnable and the loss really falls.  Compared: the contrastive loss and the contrastive
    accuracy (tf2/run.py:587-613, tf2/metrics.py:28-35) averaged over `window`-step windows after step `after` -- single
    steps of two different roundings of a chaotic SGD trajectory differ by the batch-to-batch noise, windows do not.
    Gates: |loss_bf16 / loss_f32 - 1| <= loss_rel_tol and |acc_bf16 - acc_f32| <= acc_tol in every window."""
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step
    g = torch.Generator().manual_seed(seed)
    feats, labs = [], []
    for _ in range(pool):
        feats.append(correlated_views(batch, image_size, g).to(DEV))
        labs.append({'labels': torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes).float().to(DEV)})
    curves = {}
    # 'f32r': fp32 arithmetic on inputs rounded ONCE to bf16 -- how far a single 2^-9 perturbation of the data moves the
    # same chaotic trajectory (reported as `f32_input_rounding_*`, the yardstick for the bf16 deviation; not gated)
    for mode in ('f32', 'bf16', 'f32r') if yardstick else ('f32', 'bf16'):
        FLAGS.reset()
        FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype=mode[:4].rstrip('r') if mode != 'bf16' else mode,
                     use_blur=False, train_batch_size=batch, weight_decay=1e-6)
        RT.reset()
        RT.device = torch.device(DEV)
        RT.seed = 4321                        # same initial weights in both modes
        model = model_lib.Model(num_classes)
        opt = model_lib.build_optimizer(lr)
        step_fn = make_single_step(model, opt, None)
        loss_t, acc_t = [], []
        for i in range(steps):
            f_in = feats[i % pool].bfloat16().float() if mode == 'f32r' else feats[i % pool]
            out = step_fn(f_in, labs[i % pool])
            loss_t.append(out['con_loss'].value.reshape(-1)[0].clone())
            acc_t.append(out['logits_con'].contrast_acc.reshape(-1)[0].clone())
        torch.cuda.synchronize()
        curves[mode] = (torch.stack(loss_t).double().cpu(), torch.stack(acc_t).double().cpu())
    FLAGS.reset(); RT.reset()
    lf, af = curves['f32']; lb, ab = curves['bf16']
    res = []
    tag = 'R%d %dpx b%d %d steps' % (depth, image_size, batch, steps)
    lr_, ar_ = curves['f32r'] if yardstick else curves['f32']
    worst_l = worst_a = ref_l = ref_a = 0.0
    for w0 in range(after, steps - window + 1, window):
        sl = slice(w0, w0 + window)
        worst_l = max(worst_l, abs(float(lb[sl].mean() / lf[sl].mean()) - 1.0))
        worst_a = max(worst_a, abs(float(ab[sl].mean() - af[sl].mean())))
        ref_l = max(ref_l, abs(float(lr_[sl].mean() / lf[sl].mean()) - 1.0))
        ref_a = max(ref_a, abs(float(ar_[sl].mean() - af[sl].mean())))
    fin = lambda t: bool(torch.isfinite(t).all())
    res.append(dict(name='traj_finite ' + tag, err=0.0 if (fin(lf) and fin(lb)) else 1.0, tol=0.0, scale=1.0,
                    ok=fin(lf) and fin(lb), nbad=0, numel=2 * steps))
    res.append(dict(name='traj_loss_falls_f32 ' + tag, err=float(lf[-window:].mean() / lf[:window].mean()), tol=0.9, scale=1.0,
                    ok=bool(lf[-window:].mean() < 0.9 * lf[:window].mean()), nbad=0, numel=steps,
                    first=float(lf[:window].mean()), last=float(lf[-window:].mean())))
    res.append(dict(name='traj_contrast_loss_window_rel bf16 vs f32 ' + tag, err=worst_l, tol=loss_rel_tol, scale=1.0,
                    ok=worst_l <= loss_rel_tol, nbad=0, numel=steps, f32_last=float(lf[-window:].mean()), bf16_last=float(lb[-window:].mean()),
                    f32_input_rounding_loss_rel=ref_l, f32_input_rounding_acc_abs=ref_a))
    res.append(dict(name='traj_contrast_acc_window_abs bf16 vs f32 ' + tag, err=worst_a, tol=acc_tol, scale=1.0,
                    ok=worst_a <= acc_tol, nbad=0, numel=steps, f32_last=float(af[-window:].mean()), bf16_last=float(ab[-window:].mean())))
    res[-1]['curves'] = dict(loss_f32=[round(float(x), 5) for x in lf], loss_bf16=[round(float(x), 5) for x in lb],
                             acc_f32=[round(float(x), 4) for x in af], acc_bf16=[round(float(x), 4) for x in ab])
    return res


# ------------------------------------------------------------------ two-view augmentation
def check_augment(b=6, Hs=96, Ws=128, H=64, src='uint8', strength=1.0, seed=0):
    """simclr_augment_views (crop + bicubic resize + flip, colour jitter in random order, grayscale, clip; both views) vs
    oracle/augment.py given IDENTICAL random draws (tf2/data_util.py:443-475, tf2/data.py:52-62).  Images have different
    valid sizes inside one canvas.  Also the eval path (central crop) and the range / layout invariants."""
    from oracle import augment as oa
    from simclr_amd import data_util as du
    rng = np.random.default_rng(seed)
    sizes = np.stack([rng.integers(Hs // 2, Hs + 1, b), rng.integers(Ws // 2, Ws + 1, b)], 1)
    sizes[0] = (Hs, Ws)
    if src == 'uint8':
        canvas = rng.integers(0, 256, (b, Hs, Ws, 3), dtype=np.uint8)
        # smooth structure so that bicubic overshoot and the HSV branches are all exercised
        yy, xx = np.mgrid[0:Hs, 0:Ws]
        for i in range(b):
            wave = 127 + 120 * np.sin(yy / (3.0 + i) + xx / (5.0 + 2 * i))[..., None] * np.array([1.0, 0.6, -0.8])
            canvas[i] = np.clip(0.7 * wave + 0.3 * canvas[i], 0, 255).astype(np.uint8)
        dev_src = torch.from_numpy(canvas).to(DEV)
    else:
        canvas = rng.random((b, Hs, Ws, 3)).astype(np.float32)
        dev_src = torch.from_numpy(canvas).to(DEV)
    params = du.draw_train_params(b, sizes[:, 0], sizes[:, 1], H, H, strength, rng=rng)
    params[0, 0, 5] = 1; params[0, 0, 6:10] = (1, 0, 2, 3); params[0, 0, 14] = 0      # contrast first: mean of the raw crop
    params[1, 1, 5] = 1; params[1, 1, 6:10] = (3, 2, 0, 1); params[1, 1, 14] = 1      # contrast last + grayscale
    got = du.two_view_batch(dev_src, H, H, strength, sizes=sizes, params=params)
    torch.cuda.synchronize()
    imgs = [canvas[i, :sizes[i, 0], :sizes[i, 1]] for i in range(b)]
    ref = oa.two_view_batch(imgs, params.astype(np.float64), H, H)
    g = got.double().cpu().numpy()
    err = np.abs(g - ref)
    tag = 'b%d %dx%d->%d %s s=%g' % (b, Hs, Ws, H, src, strength)
    res = [dict(name='augment_two_view_max ' + tag, err=float(err.max()), tol=5e-3, scale=1.0, ok=bool(err.max() <= 5e-3),
                nbad=int((err > 5e-3).sum()), numel=err.size),
           # a 1/1024 weight-table index may differ by one where delta*1024 sits on a rounding boundary: rare pixels
           dict(name='augment_two_view_p9999 ' + tag, err=float(np.quantile(err, 0.9999)), tol=3e-5, scale=1.0,
                ok=bool(np.quantile(err, 0.9999) <= 3e-5), nbad=int((err > 3e-5).sum()), numel=err.size),
           dict(name='augment_range_and_shape ' + tag, err=float(max(-g.min(), g.max() - 1.0, 0.0)), tol=0.0, scale=1.0,
                ok=bool(g.min() >= 0.0 and g.max() <= 1.0 and g.shape == (b, H, H, 6)), nbad=0, numel=1)]
    ev = du.preprocess_for_eval_batch(dev_src, H, H, sizes=sizes).double().cpu().numpy()
    ev_ref = np.stack([oa.preprocess_for_eval(im, H, H) for im in imgs])
    e2 = np.abs(ev - ev_ref)
    res.append(dict(name='augment_eval_center_crop ' + tag, err=float(np.quantile(e2, 0.9999)), tol=3e-5, scale=1.0,
                    ok=bool(np.quantile(e2, 0.9999) <= 3e-5 and e2.max() <= 5e-3), nbad=int((e2 > 3e-5).sum()), numel=e2.size))
    return res


# ------------------------------------------------------------------ BatchNorm backward folded into the producing 1x1 conv
def check_bn_fold(V, H, K, N, dtype, seed=0, mask_mode=2):
    """The folded form of conv(1x1, K->N) -> BatchNorm backward (csrc/bn.hip bn_fold_*, simclr_conv2d_dgrad_bn_ext) vs float64:
    with c = h W and dh = a*dm + b*c + d,  dW = h^T dh  and  d(h) = dh W^T (then the ReLU mask / sums of the producer BN of h)."""
    g = torch.Generator(device=DEV).manual_seed(seed)

    def rnd(shape, scale=1.0, shift=0.0):
        return (torch.randn(shape, device=DEV, generator=g) * scale + shift).to(dtype)
    M = V * H * H
    h = torch.relu(rnd((V, H, H, K), 1.0, 0.4))                 # activated input (non-negative, non-zero mean)
    dm = rnd((V, H, H, N), 1.0)
    w = rnd((K, N), K ** -0.5)                                   # the compute copy [Cin][Cout] = dgrad layout
    a = (0.5 + torch.rand(N, device=DEV, generator=g))
    b = 0.05 * torch.randn(N, device=DEV, generator=g)
    d = 0.1 * torch.randn(N, device=DEV, generator=g)
    bn_x = rnd((V, H, H, K), 1.5, 0.3)
    scale = torch.rand(K, device=DEV, generator=g) - 0.4
    shift = 0.3 * torch.randn(K, device=DEV, generator=g)
    mean = 0.2 * torch.randn(K, device=DEV, generator=g)
    rstd = 0.5 + torch.rand(K, device=DEV, generator=g)
    # device path (the sequence of Conv2dFixedPadding.backward_folded)
    # BN quantities that reproduce the chosen (a, b, d): a = scale, b = -scale*c2*rstd, d = scale*(c2*mean*rstd - c1)
    bscale, brstd = a, torch.ones(N, device=DEV)
    bmean = 0.3 * torch.randn(N, device=DEV, generator=g)
    c2v = -b / a
    c1v = c2v * bmean - d / a
    a2, b2, d2, wb, wext, e = ops.bn_fold_pre(w, bscale, bmean, brstd, c1v, c2v)
    a, b, d = a2, b2, d2                                            # use exactly what the kernel derived (fp32 rounding)
    w32 = w.float()
    q = ops.small_gemm_nt(wb, w32)
    t1 = ops.conv2d_wgrad(h, dm, 1, 1, 1, 0)
    gm2 = ops.conv2d_wgrad(h, h, 1, 1, 1, 0)                    # generic path (any K) ...
    ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    cs2 = ops.bn_reduce_slots(ops.bn_bwd_reduce(h, h, None, None, None, zeros, ones, 0))
    if ops.gram_supported(K, dtype):                            # ... and the single-pass Gram + column-sum kernel
        gm, cs = ops.conv2d_gram(h)
    else:
        gm, cs = gm2, cs2
    gw = ops.small_gemm_nt(gm, w32.t().contiguous())
    dw = torch.empty(K, N, device=DEV)
    ops.bn_fold_post(t1, gw, cs, a, b, d, q, dw, wext)
    bn = dict(x=bn_x, mask=None, scale=scale, shift=shift, mean=mean, rstd=rstd, mode=mask_mode)
    dmi, part = ops.conv2d_dgrad_bn_ext(dm, h, wext, e, bn)
    sums = ops.bn_reduce_slots(part)
    torch.cuda.synchronize()
    # float64 reference
    h64, dm64, w64 = h.double().view(M, K), dm.double().view(M, N), w.double()
    c = h64 @ w64
    dh = a.double() * dm64 + b.double() * c + d.double()
    dw_ref = h64.t() @ dh
    dx = dh @ w64.t()
    mk = (bn_x.double().view(M, K) * scale.double() + shift.double()) > 0
    dmi_ref = torch.where(mk, dx, torch.zeros((), device=DEV, dtype=torch.float64))
    xh = (bn_x.double().view(M, K) - mean.double()) * rstd.double()
    tag = 'V%d %dx%d %d->%d %s' % (V, H, H, K, N, str(dtype).split('.')[-1])
    bf = dtype == torch.bfloat16
    l1 = float((dmi_ref.abs().sum(0) + (dmi_ref * xh).abs().sum(0)).max())
    # sum(dm * x^) of the folded BN from T1 (no pass over c): x^ = (c - bmean) * brstd with c = h w
    sums_s2 = torch.zeros(2, N, device=DEV, dtype=torch.float64)
    sums_s2[0] = dm.double().view(M, N).sum(0)
    ops.bn_fold_s2(t1, w, bmean, brstd, sums_s2)
    torch.cuda.synchronize()
    s2_ref = (dm64 * ((c - bmean.double()) * brstd.double())).sum(0)
    gram_ref = h64.t() @ h64
    return [_res('bn_fold_colsum ' + tag, cs2[0], h64.sum(0), 1e-6),
            _res('bn_fold_colsum_gramkernel ' + tag, cs[0] if cs.dim() == 2 else cs, h64.sum(0), 2e-6),
            _res('bn_fold_gram ' + tag, gm, gram_ref, 2e-5), _res('bn_fold_gram_generic ' + tag, gm2, gram_ref, 2e-5),
            _res('bn_fold_q ' + tag, q, (w.double() * b.double()) @ w.double().t(), 2e-5),
            _res('bn_fold_s2_from_gemm ' + tag, sums_s2[1], s2_ref, 0, 2e-5 * float((dm64 * c).abs().sum(0).max())),
            _res('bn_fold_dw ' + tag, dw, dw_ref, 2e-3 if bf else 2e-4),
            _res('bn_fold_dgrad_dm ' + tag, dmi.double().view(M, K), dmi_ref, 1.5e-2 if bf else 2e-4),
            _res('bn_fold_dgrad_sum ' + tag, sums[0], dmi_ref.sum(0), 0, (2e-3 if bf else 1e-4) * l1),
            _res('bn_fold_dgrad_sumxhat ' + tag, sums[1], (dmi_ref * xh).sum(0), 0, (2e-3 if bf else 1e-4) * l1)]


def check_pool_bn_bwd_fusion(V, H, C, dtype, seed=0):
    """Stem backward with the max-pool backward fused into the BatchNorm backward (simclr_bn_bwd_reduce_pool /
    _apply_pool) vs float64 autograd of relu(x*scale+shift) -> max_pool(3, 2, SAME) (tf2/resnet.py:602-611)."""
    g = torch.Generator().manual_seed(seed)
    x = _rand((V, H, H, C), dtype, g) * 1.5 + 0.2
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    xr = x.double().requires_grad_(True)
    axes = (0, 1, 2)
    mean = xr.mean(axes)
    var = ((xr - mean) ** 2).mean(axes)
    rstd = torch.rsqrt(var + 1e-5)
    y = F.relu((xr - mean) * rstd * gamma.double() + beta.double())
    OH, pt = ops.same_pad(H, 3, 2)
    total = max((OH - 1) * 2 + 3 - H, 0)
    pr = F.max_pool2d(F.pad(y.permute(0, 3, 1, 2), (pt, total - pt, pt, total - pt), value=float('-inf')), 3, 2)
    dy = _rand((V, OH, OH, C), dtype, g)
    pr.backward(dy.double().permute(0, 3, 1, 2))
    # device: forward pool (gives arg), then the fused backward
    mean_d, rstd_d = mean.detach().float().to(DEV), rstd.detach().float().to(DEV)
    scale = (gamma.double() * rstd.detach()).float().to(DEV)
    shift = (beta.double() - mean.detach() * gamma.double() * rstd.detach()).float().to(DEV)
    xd = x.to(DEV)
    _, arg = ops.bnrelu_maxpool_fwd(xd, scale, shift)
    part = ops.bn_bwd_reduce_pool(dy.to(DEV), arg, xd, scale, shift, mean_d, rstd_d)
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    c1, c2 = ops.bn_bwd_finalize(None, None, V * H * H, dgamma, dbeta, partial=part)
    dx = ops.bn_bwd_apply_pool(dy.to(DEV), arg, xd, scale, shift, mean_d, rstd_d, c1, c2)
    # the un-fused sequence must agree with the fused one bit for bit in the gradient it would have produced
    d_un = ops.maxpool_bwd(dy.to(DEV), arg, H, H)
    part_un = ops.bn_bwd_reduce(d_un, xd, None, scale, shift, mean_d, rstd_d, 2)
    torch.cuda.synchronize()
    tag = 'V%d %d C%d %s' % (V, H, C, str(dtype).split('.')[-1])
    bf = dtype == torch.bfloat16
    # in bf16 the un-fused path rounds the un-pooled gradient to bf16 before summing; the fused one sums fp32 values
    res = [_res('poolfuse_dx ' + tag, dx, xr.grad, 1e-2 if bf else 5e-5),
           _res('poolfuse_sums_vs_unfused ' + tag, ops.bn_reduce_slots(part), ops.bn_reduce_slots(part_un), 2e-3 if bf else 1e-5)]
    if not bf and C % 32 == 0:
        # pre-split output (the stem's weight gradient reads bf16 pieces): exactly bf16(dx) and bf16(dx - hi) of the plain output
        dps = ops.bn_bwd_apply_pool(dy.to(DEV), arg, xd, scale, shift, mean_d, rstd_d, c1, c2, ps_out=True)
        torch.cuda.synchronize()
        _, hi, lo = ps_decode(dps)
        want_hi = dx.cpu().bfloat16().float()
        res += [_res('poolfuse_ps_hi ' + tag, hi, want_hi, 0.0),
                _res('poolfuse_ps_lo ' + tag, lo, (dx.cpu() - want_hi).bfloat16().float(), 0.0)]
    return res


def check_small_gemm(M, N, K, seed=0):
    """simclr_small_gemm_nt_f32 (C = A B^T, exact f32 MFMA) vs float64; both tile shapes (32 / 64) are reached by the
    sizes the folded BatchNorm backward uses (K x K x 4K and K x 4K x K, K = 64 ... 512)."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    A = torch.randn(M, K, device=DEV, generator=g)
    B = torch.randn(N, K, device=DEV, generator=g)
    C = ops.small_gemm_nt(A, B)
    ref = A.double() @ B.double().t()
    err = float((C.double() - ref).abs().max())
    scale = float(ref.abs().max())
    tol = 4e-6 * scale * max(1.0, (K / 256.0) ** 0.5)
    return [dict(name='small_gemm_nt_f32 %dx%dx%d' % (M, N, K), err=err, tol=tol, scale=scale, ok=bool(err <= tol), nbad=0, numel=M * N)]


def check_conv_fwd_bn_apply(V, H, Cin, Cout, k=1, stride=1, with_res=True, relu=True, seed=0, res_bn=False):
    """simclr_conv2d_fwd(y = NULL) + simclr_conv2d_fwd_bn_apply (the conv3 -> bn3 -> + shortcut -> relu tail of
    tf2/resnet.py:470-487 in two passes over the convolution, its output never stored) against the three-kernel path
    conv2d_fwd -> bn_finalize -> bn_apply: statistics, output and ReLU bit mask must be BIT-IDENTICAL."""
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = (k - 1) // 2
    OH = (H + (k - 1) - k) // stride + 1
    x = (torch.randn(V, H, H, Cin, device=DEV, generator=g)).to(dtype)
    w = torch.randn(k, k, Cin, Cout, device=DEV, generator=g) * (k * k * Cin) ** -0.5
    res = torch.randn(V, OH, OH, Cout, device=DEV, generator=g).to(dtype) if with_res else None
    gamma = torch.rand(Cout, device=DEV, generator=g) + 0.5
    beta = 0.2 * torch.randn(Cout, device=DEV, generator=g)
    w_t = ops.prep_weights(w, 0, dtype)
    M = V * OH * OH
    # reference path
    st_a = ops.conv_stats(M, Cout, DEV)
    c = ops.conv2d_fwd(x, w_t, k, k, stride, pad, OH, OH, stats=st_a)
    mm, mv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean, rstd, scale, shift = ops.bn_finalize(None, M, gamma, beta, mm, mv, 0.9, partial=st_a)
    rs = (torch.rand(Cout, device=DEV, generator=g) + 0.5) if res_bn else None
    rb = (0.3 * torch.randn(Cout, device=DEV, generator=g)) if res_bn else None
    y_ref, bits_ref = (ops.bn_apply(c, scale, shift, relu, res=res, rscale=rs, rshift=rb, want_bits=True) if relu
                       else (ops.bn_apply(c, scale, shift, relu, res=res, rscale=rs, rshift=rb), None))
    # fused path
    st_b = ops.conv_stats(M, Cout, DEV)
    ops.conv2d_fwd(x, w_t, k, k, stride, pad, OH, OH, stats=st_b, store=False)
    mm2, mv2 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    _, _, scale2, shift2 = ops.bn_finalize(None, M, gamma, beta, mm2, mv2, 0.9, partial=st_b)
    out = ops.conv2d_fwd_bn_apply(x, w_t, k, k, stride, pad, OH, OH, scale2, shift2, res=res, relu=relu, want_bits=relu,
                                  rscale=rs, rshift=rb)
    y, bits = out if relu else (out, None)
    torch.cuda.synchronize()
    tag = 'V%d %dx%d %d->%d k%d s%d res%d relu%d resbn%d' % (V, H, H, Cin, Cout, k, stride, int(with_res), int(relu), int(res_bn))

    def same(name, a, b):
        ok = bool(torch.equal(a, b))
        nbad = 0 if ok else int((a != b).sum())
        return dict(name='fwd_bn_apply_%s %s' % (name, tag), err=float(nbad), tol=0.0, scale=0.0, ok=ok, nbad=nbad, numel=a.numel())
    res_l = [same('stats', ops.bn_reduce_slots(st_a), ops.bn_reduce_slots(st_b)), same('scale', scale, scale2),
             same('y', y_ref.view(torch.int16), y.view(torch.int16))]
    if relu:
        res_l.append(same('bits', bits_ref, bits))
    # and the output is right in the first place: float64 reference of relu(bn(conv) + res) from the bf16-rounded conv
    rterm = 0.0
    if with_res:
        rterm = res.double() * rs.double() + rb.double() if res_bn else res.double()
    ref = c.double() * scale.double() + shift.double() + rterm
    if relu:
        ref = ref.clamp_min(0.0)
    err = float((y.double() - ref).abs().max())
    sc_ = float(ref.abs().max())
    res_l.append(dict(name='fwd_bn_apply_value ' + tag, err=err, tol=2.0 ** -7 * sc_, scale=sc_, ok=bool(err <= 2.0 ** -7 * sc_), nbad=0, numel=y.numel()))
    return res_l


def check_conv_fwd_bn_apply_f32(V, H, Cin, Cout, matmul='f16x3_3', with_res=True, relu=True, res_bn=False, seed=0):
    """Round 6: the fused bottleneck tail in fp32 storage -- simclr_conv2d_fwd_bn_apply(SIMCLR_DT_F32) applies bn_apply's arithmetic to
    the fp32 accumulators in the convolution's epilogue: output and ReLU bit mask (one byte per 4 channels) BIT-IDENTICAL to
    conv2d_fwd -> bn_apply in the same matrix arithmetic."""
    ops.set_f32_matmul(matmul)
    try:
        g = torch.Generator(device=DEV).manual_seed(seed)
        x = torch.randn(V, H, H, Cin, device=DEV, generator=g)
        w = torch.randn(1, 1, Cin, Cout, device=DEV, generator=g) * Cin ** -0.5
        res = torch.randn(V, H, H, Cout, device=DEV, generator=g) if with_res else None
        gamma = torch.rand(Cout, device=DEV, generator=g) + 0.5
        beta = 0.2 * torch.randn(Cout, device=DEV, generator=g)
        w_t = ops.prep_weights(w, 0, torch.float32)
        M = V * H * H
        st = ops.conv_stats(M, Cout, DEV)
        c = ops.conv2d_fwd(x, w_t, 1, 1, 1, 0, H, H, stats=st)
        mm, mv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        _, _, scale, shift = ops.bn_finalize(None, M, gamma, beta, mm, mv, 0.9, partial=st)
        rs = (torch.rand(Cout, device=DEV, generator=g) + 0.5) if res_bn else None
        rb = (0.3 * torch.randn(Cout, device=DEV, generator=g)) if res_bn else None
        y_ref, bits_ref = (ops.bn_apply(c, scale, shift, relu, res=res, rscale=rs, rshift=rb, want_bits=True) if relu
                           else (ops.bn_apply(c, scale, shift, relu, res=res, rscale=rs, rshift=rb), None))
        out = ops.conv2d_fwd_bn_apply(x, w_t, 1, 1, 1, 0, H, H, scale, shift, res=res, relu=relu, want_bits=relu, rscale=rs, rshift=rb)
        y, bits = out if relu else (out, None)
        torch.cuda.synchronize()
        tag = 'V%d %dx%d %d->%d f32/%s res%d relu%d resbn%d' % (V, H, H, Cin, Cout, matmul, int(with_res), int(relu), int(res_bn))

        def same(name, a, b):
            ok = bool(torch.equal(a, b))
            nbad = 0 if ok else int((a != b).sum())
            return dict(name='fwd_bn_apply_f32_%s %s' % (name, tag), err=float(nbad), tol=0.0, scale=0.0, ok=ok, nbad=nbad, numel=a.numel())
        out_l = [same('y', y_ref.view(torch.int32), y.view(torch.int32))]
        if relu:
            out_l.append(same('bits', bits_ref.reshape(-1), bits.reshape(-1)))
        return out_l
    finally:
        ops.set_f32_matmul('exact')


def check_conv_pivoted_stats(V, H, Cin, Cout, k, stride, offset=300.0, seed=0, matmul='exact'):
    """fp32 convolution whose output has |mean| >> sigma per channel (input = offset + noise): BatchNorm mean / variance from
    simclr_conv2d_fwd_pivoted + simclr_bn_reduce_slots_pivoted + simclr_bn_finalize against float64 moments OF THE STORED OUTPUT
    (isolates the statistics from the convolution's own rounding).  Raw fp32 moments (SIMCLR_BN_PIVOT=0) lose about
    (mean / sigma)^2 * 2^-24 of the variance: reported next to the pivoted error, not gated."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = (k - 1) // 2
    OH = (H + (k - 1) - k) // stride + 1
    x = offset + torch.randn(V, H, H, Cin, device=DEV, generator=g)
    w = torch.randn(k, k, Cin, Cout, device=DEV, generator=g) * (k * k * Cin) ** -0.5
    w_t = ops.prep_weights(w, 0, torch.float32)
    M = V * OH * OH
    gamma, beta = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    ops.set_f32_matmul(matmul)
    out = {}
    try:
        for mode in ('1', '0'):
            os.environ['SIMCLR_BN_PIVOT'] = mode
            stats = ops.conv_stats(M, Cout, DEV)
            y, st, sums = ops.conv2d_fwd_with_stats(x, w_t, k, k, stride, pad, OH, OH, stats)
            assert (sums is not None) == (mode == '1')
            mm, mv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
            mean, rstd, _, _ = ops.bn_finalize(sums, M, gamma, beta, mm, mv, 0.9, partial=st)
            torch.cuda.synchronize()
            y64 = y.double().view(M, Cout)
            m64 = y64.mean(0)
            v64 = (y64 - m64).pow(2).mean(0)
            var = rstd.double().pow(-2) - 1e-5
            out[mode] = (float(((mean.double() - m64).abs() / v64.sqrt()).max()), float(((var - v64).abs() / v64).max()),
                         float((m64.abs() / v64.sqrt()).median()), float((m64.abs() / v64.sqrt()).max()))
    finally:
        os.environ.pop('SIMCLR_BN_PIVOT', None)
        ops.set_f32_matmul('exact')
    tag = 'V%d %dx%d %d->%d k%d s%d offset %g %s (median |mean|/sigma %.0f)' % (V, H, H, Cin, Cout, k, stride, offset, matmul, out['1'][2])
    # the mean leaves simclr_bn_finalize as a FLOAT: half an ulp of |mean| = 2^-24 |mean| / sigma in units of sigma (both paths alike)
    tol_m = 2e-6 + 2.0 ** -23 * out['1'][3]
    return [dict(name='pivoted_bn_mean_over_sigma ' + tag, err=out['1'][0], tol=tol_m, scale=1.0, ok=bool(out['1'][0] <= tol_m), nbad=0,
                 numel=Cout, raw_moments_err=out['0'][0]),
            dict(name='pivoted_bn_var_rel ' + tag, err=out['1'][1], tol=2e-5, scale=1.0, ok=bool(out['1'][1] <= 2e-5), nbad=0,
                 numel=Cout, raw_moments_err=out['0'][1])]


def check_gram_stats(V, H, K, N, seed=0):
    """BatchNorm statistics of c = h W from the Gram matrix of h (simclr_conv2d_gram -> simclr_small_gemm_nt_f32 ->
    simclr_bn_sums_from_gram) against (a) float64 sums of the exact products and (b) the statistics the convolution
    epilogue accumulates.  h is a post-ReLU-like activation (non-negative, mean ~ 0.6 sigma: the cancellation-prone case)."""
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(seed)
    h = torch.relu(torch.randn(V, H, H, K, device=DEV, generator=g) + 0.3).to(dtype)
    w = torch.randn(1, 1, K, N, device=DEV, generator=g) * K ** -0.5
    w_t = ops.prep_weights(w, 0, dtype)                      # [N, K] bf16
    w_d = ops.prep_weights(w, 1, dtype)                      # [K, N] bf16
    M = V * H * H
    if ops.gram_supported(K, dtype):
        gm, cs = ops.conv2d_gram(h)
    else:
        gm = ops.conv2d_wgrad(h, h, 1, 1, 1, 0)
        ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        cs = ops.bn_reduce_slots(ops.bn_bwd_reduce(h, h, None, None, None, zeros, ones, 0))
    gw = ops.small_gemm_nt(gm, w_t.float())
    sums = ops.bn_sums_from_gram(gw, w_d.float(), cs)
    st = ops.conv_stats(M, N, DEV)
    ops.conv2d_fwd(h, w_t, 1, 1, 1, 0, H, H, stats=st)
    sums_conv = ops.bn_reduce_slots(st)
    torch.cuda.synchronize()
    c64 = h.reshape(M, K).double() @ w_d.double()
    ref = torch.stack([c64.sum(0), (c64 * c64).sum(0)])
    mean_ref = ref[0] / M
    var_ref = ref[1] / M - mean_ref ** 2
    out = []
    for name, sm in (('gram', sums), ('conv_epilogue', sums_conv)):
        mean = sm[0] / M
        var = sm[1] / M - mean ** 2
        sd = var_ref.sqrt()
        e_mean = float(((mean - mean_ref).abs() / sd).max())           # in units of the channel's standard deviation
        e_var = float(((var - var_ref).abs() / var_ref).max())
        tag = '%s V%d %dx%d %d->%d' % (name, V, H, H, K, N)
        # the wide (256 x 256) forward tile takes its statistics in the row-wise pass, i.e. of the bf16-ROUNDED outputs (the
        # tensor the reference's moments see): against the exact products that adds unbiased rounding noise at the 1e-5 sigma level
        t_mean = 2e-5 if name == 'gram' else 5e-5
        out.append(dict(name='bn_stats_mean ' + tag, err=e_mean, tol=t_mean, scale=1.0, ok=e_mean <= t_mean, nbad=0, numel=N))
        out.append(dict(name='bn_stats_var ' + tag, err=e_var, tol=1e-4, scale=1.0, ok=e_var <= 1e-4, nbad=0, numel=N))
    return out


# ------------------------------------------------------------------ the PRODUCT against the reference-source fixtures, in one hop
_PIN = {}


def reference_pin():
    """(make_reference_golden module, fixtures): tests/golden/reference_pin.npz = outputs of /root/reference/tf2/*.py themselves."""
    if not _PIN:
        import importlib.util
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        spec = importlib.util.spec_from_file_location('make_reference_golden', os.path.join(here, 'golden', 'make_reference_golden.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _PIN['m'], _PIN['ref'] = m, dict(np.load(m.OUT_NPZ))
    return _PIN['m'], _PIN['ref']


def pinned_product_model(mm, compute_dtype='f32', f32_matmul='exact', weight_decay=None, strategy=None, variables=None):
    """simclr_amd.model.Model under the flags of fixture case `mm`, with the case's variables injected by NAME.
    `variables`: name -> float64 array (default: the table make_reference_golden.py handed to the reference's own model)."""
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    m, _ = reference_pin()
    FLAGS.reset()
    FLAGS.update(use_blur=False, resnet_depth=mm['depth'], image_size=mm['size'], sk_ratio=mm['sk'], compute_dtype=compute_dtype,
                 f32_matmul=f32_matmul, train_batch_size=mm['batch'],
                 **{flag: mm[k] for k, flag in m.MODEL_FLAGS.items() if k in mm})
    if weight_decay is not None:
        FLAGS.update(weight_decay=weight_decay)
    RT.reset()
    RT.device = torch.device(DEV)
    RT.strategy = strategy
    model = model_lib.Model(mm['classes'])
    with torch.no_grad():
        model(torch.zeros(2, mm['size'], mm['size'], 6, device=DEV), training=False)      # builds the variables; nothing moves in inference mode
    if variables is None:
        variables = _pin_variables(mm)
    names = sorted(v.name for v in model.variables)
    assert names == sorted(variables), (sorted(set(names) - set(variables))[:4], sorted(set(variables) - set(names))[:4])
    if mm.get('recipe'):
        # the `*_img` variables are a function of (reference name, shape, the owner's initial value): derived from the PRODUCT's own freshly
        # initialised variables they must equal the table the reference's model was given (bench.py's in-run parity block relies on it)
        import recipe
        for v in model.variables:
            mine = recipe.variable_value(v.name[len('model/'):], v.value.double().cpu().numpy(), mm['perturb'])
            assert np.array_equal(mine, np.asarray(variables[v.name])), v.name
    for v in model.variables:
        assert tuple(v.value.shape) == tuple(variables[v.name].shape), (v.name, v.value.shape, variables[v.name].shape)
        v.value.copy_(torch.from_numpy(np.asarray(variables[v.name])).to(torch.float32).to(DEV))
    RT.weights_version += 1
    return model


def _pin_variables(mm):
    """name -> float64 array: the variables make_reference_golden.py injected into the reference's model for case `mm` (cached)"""
    key = ('vars', mm['tag'])
    if key not in _PIN:
        m, _ = reference_pin()
        _, params, state, _ = m._oracle_model(mm)
        _PIN[key] = {k: v.numpy() for k, v in list(params.items()) + list(state.items())}
    return _PIN[key]


def _l2n(x):
    x = np.asarray(x, dtype=np.float64)
    return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))       # tf.math.l2_normalize (tf2/objective.py:54)


def _pin_fp32_noise(mm):
    """What fp32 arithmetic itself does to this case: the oracle evaluated in torch-CPU float32 against the float64 fixtures
    (normalised-embedding abs error, projection rel error, linear-eval logits rel error).  The small wiring cases (batch 3 ... 4:
    BatchNorm over 6 ... 8 rows in the heads) sit at 5e-5 ... 1e-4 here -- their gates are calibrated by this number; the `*_img`
    cases sit at 1e-6 ... 2e-6 and are gated by north_star's fixed 1e-5 / 1e-3."""
    from collections import OrderedDict
    from oracle.model_torch import Builder
    m, ref = reference_pin()
    if ('noise', mm['tag']) in _PIN:
        return _PIN[('noise', mm['tag'])]
    cfg, params, state, _ = m._oracle_model(mm)
    images, _ = m._model_inputs(mm)
    b = Builder(cfg, params=OrderedDict((k, v.float()) for k, v in params.items()),
                state=OrderedDict((k, v.float()) for k, v in state.items()), dtype=torch.float32)
    with torch.no_grad():
        proj, sup = b.model(torch.from_numpy(images).float(), training=True)
        # the inference forward on the moving statistics that training forward left behind: calibrated SEPARATELY -- one update of
        # the moving averages (0.9 x initial + 0.1 x batch) does not normalise the activations, and the deep SK model then grows to
        # |proj| ~ 1e4 with an fp32 noise of 1.6e-3 on the embeddings (fp64 arithmetic on fp32-rounded inputs alone: 1.4e-4)
        b2 = Builder(cfg, params=b.params, state=OrderedDict((k, v) for k, v in b.new_state.items()), dtype=torch.float32)
        proj_e, sup_e = b2.model(torch.from_numpy(images).float(), training=False)
    t = mm['tag']
    out = {}
    for sfx, pj, sp in (('', proj, sup), ('_eval', proj_e, sup_e)):
        r = ref[t + '_proj' + sfx]
        out['emb' + sfx] = float(np.abs(_l2n(pj.double().numpy()) - _l2n(r)).max())
        out['proj' + sfx] = float(np.abs(pj.double().numpy() - r).max() / np.abs(r).max())
        if sp is not None:
            out['sup' + sfx] = float(np.abs(sp.double().numpy() - ref[t + '_sup' + sfx]).max() / np.abs(ref[t + '_sup' + sfx]).max())
    _PIN[('noise', mm['tag'])] = out
    return out


def check_reference_pin_model(tag, compute_dtype='f32', f32_matmul='exact', gate=True):
    """simclr_amd.model.Model (train forward, BatchNorm moving statistics, inference forward, add_weight_decay) against what
    /root/reference/tf2/model.py:228-280 ITSELF returned on the same variables and images (tests/golden/reference_pin.npz) -- the
    product against the reference's source in ONE hop, no oracle in between.  fp32 modes: normalised embeddings <= 1e-5 abs on the
    well-conditioned `*_img` cases (north_star); on the small wiring cases the gate is max(1e-5, 6 x the error torch-CPU fp32 shows)."""
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    m, ref = reference_pin()
    mm = next(c for c in m.MODELS if c['tag'] == tag)
    images, _ = m._model_inputs(mm)
    noise = _pin_fp32_noise(mm) if gate else dict(emb=0.0, proj=0.0, sup=0.0, emb_eval=0.0, proj_eval=0.0, sup_eval=0.0)
    fixed = bool(mm.get('recipe'))
    model = pinned_product_model(mm, compute_dtype, f32_matmul)
    x = torch.from_numpy(images).float().to(DEV)
    mode = compute_dtype + ('' if f32_matmul == 'exact' else '/' + f32_matmul)
    res = []

    def entry(name, err, tol, **kw):
        d = dict(name='pin_%s %s %s' % (name, tag, mode), err=float(err), tol=float(tol) if gate else float('inf'), scale=1.0,
                 ok=bool(err <= tol) or not gate, nbad=0, numel=1)
        d.update(kw)
        res.append(d)

    def rel(a, r):
        return float(np.abs(np.asarray(a, dtype=np.float64) - r).max() / (np.abs(r).max() + 1e-300))

    for training, sfx in ((True, ''), (False, '_eval')):
        emb_tol = 1e-5 if fixed else max(1e-5, 6.0 * noise['emb' + sfx])
        rel_tol = 5e-5 if fixed else max(5e-5, 6.0 * max(noise['proj' + sfx], noise.get('sup' + sfx, 0.0)))
        proj, sup = model(x, training=training)
        torch.cuda.synchronize()
        p = proj.double().cpu().numpy()
        entry('embeddings_abs' + sfx, np.abs(_l2n(p) - _l2n(ref[tag + '_proj' + sfx])).max(), emb_tol, fp32_noise=noise['emb' + sfx])
        entry('proj_rel' + sfx, rel(p, ref[tag + '_proj' + sfx]), rel_tol)
        if ref[tag + '_sup' + sfx].size:
            entry('sup_logits_rel' + sfx, rel(sup.dense().double().cpu().numpy(), ref[tag + '_sup' + sfx]), rel_tol)
        else:
            entry('no_sup_head' + sfx, 0.0 if sup is None else 1.0, 0.0)
        if training:      # the moving averages after ONE training forward (tf2/resnet.py:62-72, batch_norm_decay 0.9), per variable, by name
            mv = sorted((v.name, v.value.double().cpu().numpy()) for v in model.variables if 'moving_' in v.name)
            got = np.array([[float(a.sum()), float(np.abs(a).sum())] for _, a in mv])
            want = ref[tag + '_moving_checksum']
            entry('moving_statistics_checksum_rel', np.abs(got - want).max() / np.abs(want).max() if got.shape == want.shape else float('inf'),
                  2e-5 if fixed else max(2e-5, 6.0 * noise['proj']), n=len(mv))
    FLAGS.update(weight_decay=1e-4)
    wd = model_lib.add_weight_decay(model, adjust_per_optimizer=True)            # tf2/model.py:47-60
    wd = float(wd.reshape(-1)[0]) if torch.is_tensor(wd) else float(wd)
    entry('weight_decay_lars_rel', abs(wd - float(ref[tag + '_wd_lars'])) / max(abs(float(ref[tag + '_wd_lars'])), 1e-30)
          if float(ref[tag + '_wd_lars']) != 0.0 else abs(wd), 2e-6, value=wd, ref=float(ref[tag + '_wd_lars']))
    FLAGS.reset(); RT.reset()
    return res


def check_reference_pin_step(tag, compute_dtype='f32', f32_matmul='exact', gate=True):
    """simclr_amd.run.make_single_step against the scaled loss and the seven metrics tf2/run.py:557-622 ITSELF produced (one replica;
    the two-replica fixtures are checked in tests/test_gpu_distributed.py).  Loss terms <= 1e-3 relative (north_star); the
    normalised embeddings the loss was computed from <= 1e-5 on the `*_img` cases."""
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step
    m, ref = reference_pin()
    mm = next(c for c in m.MODELS if c['tag'] == tag)
    images, labels = m._model_inputs(mm)
    fixed = bool(mm.get('recipe'))
    model = pinned_product_model(mm, compute_dtype, f32_matmul, weight_decay=1e-4)
    step = make_single_step(model, model_lib.build_optimizer(0.1), None)
    out = step(torch.from_numpy(images).float().to(DEV), {'labels': torch.from_numpy(labels).float().to(DEV)})
    torch.cuda.synchronize()
    mode = compute_dtype + ('' if f32_matmul == 'exact' else '/' + f32_matmul)
    key = 'step_%s_R1' % tag
    want = dict(zip(m.STEP_METRICS, ref[key + '_metrics']))
    got = {k.split('/', 1)[1]: float(v.result()) for k, v in step.metrics.items()}
    res = []

    def entry(name, err, tol, **kw):
        d = dict(name='pin_step_%s %s %s' % (name, tag, mode), err=float(err), tol=float(tol) if gate else float('inf'), scale=1.0,
                 ok=bool(err <= tol) or not gate, nbad=0, numel=1)
        d.update(kw)
        res.append(d)
    n = images.shape[0]
    total = float(out['total_loss'].reshape(-1)[0])
    entry('scaled_loss_rel', abs(total - float(ref[key + '_scaled_loss'][0])) / abs(float(ref[key + '_scaled_loss'][0])), 1e-3,
          value=total, ref=float(ref[key + '_scaled_loss'][0]))
    for k in ('contrast_loss', 'supervised_loss', 'total_loss', 'weight_decay'):
        entry(k + '_rel', abs(got[k] - want[k]) / max(abs(want[k]), 1e-30), 1e-3 if k != 'weight_decay' else 2e-6, value=got[k], ref=want[k])
    # argmax metrics: exact on the well-conditioned cases; on the small wiring cases one near-tie may fall the other way in fp32
    entry('contrast_acc_abs', abs(got['contrast_acc'] - want['contrast_acc']), 1e-6 if fixed else 1.0 / n + 1e-6)
    entry('supervised_acc_abs', abs(got['supervised_acc'] - want['supervised_acc']), 1e-6 if fixed else 0.5 / n + 1e-6)
    entry('contrast_entropy_abs', abs(got['contrast_entropy'] - want['contrast_entropy']), 1e-3)
    if fixed:      # the embeddings the contrastive loss was computed from = l2-normalised `proj` of the training forward
        z = out['con_loss'].normalized.double().cpu().numpy()
        entry('embeddings_abs', np.abs(z - _l2n(ref[tag + '_proj'])).max(), 1e-5)
    applied = sorted(v.name[len('model/'):] for v in model._flat_order)
    entry('applied_variable_names', 0.0 if applied == list(ref[key + '_applied_names']) else 1.0, 0.0)
    if key + '_grad_fd' in ref:
        # BACKWARD values against the reference: <d loss / d variable, direction> for unit directions over whole tensors, by central differences of the reference's own
        # single_step (make_reference_golden.py GRAD_FD_VARS; the oracle's autograd agrees with them to 3e-6) vs the product's hand-written
        # backward (the gradients the step handed to LARS)
        byname = {v.name: v for v in model._flat_order}
        dirs = m.grad_fd_directions(tag, {n[len('model/'):]: tuple(v.value.shape) for n, v in byname.items()})
        want_g = ref[key + '_grad_fd']
        got_g = np.array([float((byname['model/' + n].grad.double().cpu().numpy() * d).sum()) for n, d in dirs])
        gnorm = np.array([float(byname['model/' + n].grad.double().norm()) for n, _ in dirs])
        # <g, d> for a unit direction: the error is measured against the gradient tensor's own norm (a direction nearly orthogonal to
        # the gradient says nothing relative to its own small value)
        rtol = 2e-3 if f32_matmul == 'exact' else 5e-3          # three-term backward arithmetic: ~2^-17 per product
        if not fixed:
            rtol = 5e-3      # the small ResNet-50 wiring case (batch 3): torch-CPU fp32 autograd is 1.5e-4 of the tensor norm off on these directions
        worst = float(np.max(np.abs(got_g - want_g) / (rtol * gnorm + 1e-9)))
        entry('backward_vs_reference_central_differences', worst, 1.0 if compute_dtype == 'f32' else float('inf'), n=len(dirs),
              got=[float('%.6g' % x) for x in got_g], want=[float('%.6g' % x) for x in want_g], gnorm=[float('%.4g' % x) for x in gnorm],
              worst_rel=float(np.max(np.abs(got_g - want_g) / np.abs(want_g))))
    FLAGS.reset(); RT.reset()
    return res
