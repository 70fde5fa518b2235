"""Thin torch-tensor wrappers over the C ABI (include/simclr_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function
below only validates shapes and forwards raw device pointers to libsimclr_hip.so.
All tensors must be CUDA (ROCm) tensors, contiguous.
"""
import ctypes
import os

import torch

from ._lib import DT_BF16, DT_F32, lib

NSLOT = 32  # default slot count of hand-filled statistics buffers (tests); producers ask the library (conv_stats_slots ...)


class KernelProfiler:
    """Live per-launch timing with HIP events on the launch stream (torch's current stream is the
    stream every kernel here is enqueued on).  Used by bench.py for the roofline object."""

    def __init__(self):
        self.records = []   # (family, flops, min bytes, as-implemented bytes, start_event, end_event)

    def launch(self, family, flops, nbytes, impl_bytes, fn):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.records.append((family, flops, nbytes, impl_bytes, a, b))
        return r

    def summary(self):
        out = {}
        for fam, fl, by, ib, a, b in self.records:
            d = out.setdefault(fam, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, impl_bytes=0.0))
            d['launches'] += 1
            d['ms'] += a.elapsed_time(b)
            d['flops'] += fl
            d['bytes'] += by
            d['impl_bytes'] += ib
        return out


PROFILER = None


def _launch(family, flops, nbytes, fn, impl_bytes=None):
    """nbytes: SURVEY 8(d)'s MINIMUM bytes of the op (every operand read once, the result written once);
    impl_bytes: what this implementation moves by design (extra operands of fused epilogues), default = nbytes."""
    if PROFILER is None:
        return fn()
    return PROFILER.launch(family, flops, nbytes, nbytes if impl_bytes is None else impl_bytes, fn)


def dt(t):
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError('unsupported dtype %s' % t.dtype)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'need contiguous device tensor'
    assert getattr(t, '_ps', None) is None, 'a pre-split tensor reached an operation that reads plain floats'
    return ctypes.c_void_p(t.data_ptr())


# ---- pre-split block format (csrc/common.h): an fp32-typed tensor whose 128-byte blocks hold (hi, lo) 16-bit pieces -----------
# Tensors in that format carry the Python attribute `_ps` ('b16' | 'f16'); only the operations that name it accept them (_pp).
FMT_PS_IN, FMT_PS_OUT, FMT_PS_F16, FMT_PS_IN2, FMT_PS_W = 0x100, 0x200, 0x400, 0x800, 0x100000


def ps_kind(t):
    return getattr(t, '_ps', None) if t is not None else None


def _pp(t):
    """Pointer of a tensor that may be pre-split (the callee is told through the dtype flags)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'need contiguous device tensor'
    return ctypes.c_void_p(t.data_ptr())


def _pw(w, fwd):
    """(pointer, dtype flag) of a compute-weight operand: its pre-split copy (WeightPairBatch) with SIMCLR_FMT_PS_W when one exists for the
    terms this call runs with, else the fp32 matrix itself."""
    ps = getattr(w, '_psw', None)
    if ps is not None and ps[1] == (_TERMS[0] if fwd else _TERMS[1]):
        return ctypes.c_void_p(ps[0].data_ptr()), FMT_PS_W
    return _p(w), 0


def ps_backward_enabled():
    """Gradient tensors between a BatchNorm backward and the convolution in front of it in the pre-split format: fp32 storage,
    three bf16 backward terms (FLAGS.f32_matmul in 'bf16x3' | 'bf16x6_3' | 'f16x3_3'), SIMCLR_PS_BWD != 0."""
    import os
    if os.environ.get('SIMCLR_PS_BWD', '1') in ('', '0') or os.environ.get('SIMCLR_F32_PRESPLIT', '1') in ('', '0'):
        return False
    return _TERMS[1] == 3


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------- NT-Xent
def l2norm_fwd(x):
    rows, D = x.shape
    z = torch.empty_like(x)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    _launch('l2norm_fwd', 3.0 * rows * D, 8.0 * rows * D, lambda: lib().l2norm_fwd(_p(x), _p(z), _p(inv), rows, D, _s()))
    return z, inv


def l2norm_bwd(z, inv, dz):
    dx = torch.empty_like(z)
    _launch('l2norm_bwd', 5.0 * z.numel(), 12.0 * z.numel(),
            lambda: lib().l2norm_bwd(_p(z), _p(inv), _p(dz), _p(dx), z.shape[0], z.shape[1], _s()))
    return dx


_NTXENT_DIMS = (64, 128, 256)       # embedding widths the fused kernels are instantiated for


def _ntxent_dim(D):
    """Kernel width for a `proj_out_dim` of D (a free flag in the reference, tf2/run.py:196): the next instantiated width.
    Zero columns change neither the dot products nor the norms, so other widths run zero-padded."""
    for d in _NTXENT_DIMS:
        if D <= d:
            return d
    raise ValueError('NT-Xent kernels support proj_out_dim <= %d (got %d)' % (_NTXENT_DIMS[-1], D))


def _ntxent_pad(z):
    D = z.shape[1]
    Dk = _ntxent_dim(D)
    return z if Dk == D else torch.nn.functional.pad(z, (0, Dk - D))


def ntxent_workspace(n, N, D, device):
    D = _ntxent_dim(D)
    nbytes = lib().ntxent_workspace_bytes(n, N, D)
    return torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)


def ntxent_fwd(z_local, z_all, rank, temperature, ws=None, split=False):
    """split: three fp16-piece MFMA terms per fp32 product in the sweeps (SIMCLR_FMT_TERMS(13) in the D argument) -- for
    l2-normalised rows; default = exact fp32-input MFMA."""
    n, D = z_local.shape[0] // 2, z_local.shape[1]
    N = z_all.shape[0] // 2
    assert z_local.dtype == torch.float32 and z_all.dtype == torch.float32
    z_local, z_all = _ntxent_pad(z_local), _ntxent_pad(z_all)
    D = z_local.shape[1]
    if ws is None:
        ws = ntxent_workspace(n, N, D, z_local.device)
    out = step_scalars(4, z_local.device)
    row_stats = torch.empty(2 * n, 2, device=z_local.device, dtype=torch.float32)
    _launch('ntxent_fwd', 8.0 * n * N * D, 4.0 * (2 * n + 2 * N) * D,
            lambda: lib().ntxent_fwd(_p(z_local), _p(z_all), n, N, D | ((14 << 12) if split else 0), rank, float(temperature), _p(out), _p(row_stats),
                                     _p(ws), _s()))
    return out, row_stats, ws


def ntxent_bwd(z_local, z_all, rank, temperature, row_stats, grad_scale, out, ws, split=False):
    n, D0 = z_local.shape[0] // 2, z_local.shape[1]
    N = z_all.shape[0] // 2
    z_local, z_all = _ntxent_pad(z_local), _ntxent_pad(z_all)
    D = z_local.shape[1]
    dz_local = torch.empty_like(z_local)
    dz_all = torch.empty_like(z_all)
    # bytes: the fused forward+backward I/O of SURVEY 8(d): read h_local + h_all, write dH_local + dH_all
    _launch('ntxent_bwd', 16.0 * n * N * D, 2.0 * (2 * n + 2 * N) * D * 4,
            lambda: lib().ntxent_bwd(_p(z_local), _p(z_all), n, N, D | ((14 << 12) if split else 0), rank, float(temperature), _p(row_stats),
                                     float(grad_scale), _p(dz_local), _p(dz_all), _p(out), _p(ws), _s()))
    if D != D0:
        dz_local, dz_all = dz_local[:, :D0].contiguous(), dz_all[:, :D0].contiguous()
    return dz_local, dz_all


def ntxent_logits_ab(z_local, z_all, temperature):
    z_local, z_all = _ntxent_pad(z_local), _ntxent_pad(z_all)
    n, D = z_local.shape[0] // 2, z_local.shape[1]
    N = z_all.shape[0] // 2
    out = torch.empty(n, N, device=z_local.device, dtype=torch.float32)
    lib().ntxent_logits_ab(_p(z_local), _p(z_all), n, N, D, float(temperature), _p(out), _s())
    return out


# ---------------------------------------------------------------- conv / dense
def prep_weights(w_hwio, mode, dtype, khp=0, kwp=0, out=None, cin_p=0, cout_p=0):
    """fp32 HWIO master -> compute copy.  cin_p / cout_p: zero-padded channel dims of the copy."""
    KH, KW, CI, CO = w_hwio.shape
    cip, cop = cin_p or CI, cout_p or CO
    if mode == 2:
        shape = (cop, khp * kwp * 4)
    elif mode == 0:
        shape = (cop, KH * KW * cip)
    else:
        shape = (cip, KH * KW * cop)
    if out is None:
        out = torch.empty(shape, device=w_hwio.device, dtype=dtype)
    lib().prep_weights(_p(w_hwio), _p(out), KH, KW, CI, CO, mode, khp, kwp, cip, cop, dt(out), _s())
    return out


def prep_weights_pair(w_hwio, dtype, cin_p=0, cout_p=0):
    """(w_t [cop, KH*KW*cip], w_d [cip, KH*KW*cop]) -- prep_weights modes 0 and 1 in one launch."""
    KH, KW, CI, CO = w_hwio.shape
    cip, cop = cin_p or CI, cout_p or CO
    w_t = torch.empty(cop, KH * KW * cip, device=w_hwio.device, dtype=dtype)
    w_d = torch.empty(cip, KH * KW * cop, device=w_hwio.device, dtype=dtype)
    lib().prep_weights_pair(_p(w_hwio), _p(w_t), _p(w_d), KH, KW, CI, CO, cip, cop, dt(w_t), _s())
    return w_t, w_d


class WeightPairBatch:
    """Refreshes the (w_t, w_d) compute copies of many convolutions with ONE launch (simclr_prep_weights_pair_multi).
    entries: list of (w_hwio fp32 [KH,KW,CI,CO], cin_p, cout_p); the output buffers are allocated once and keep their
    addresses, so the descriptor table is built once per (dtype, set of master pointers)."""

    def __init__(self, entries, dtype):
        dev = entries[0][0].device
        self.dtype = dtype
        self.key = tuple(w.data_ptr() for w, _, _ in entries)
        self.pairs = []
        chunk = lib().prep_chunk_elems()
        table, chunks = [], []
        for t, (w, cip, cop) in enumerate(entries):
            KH, KW, CI, CO = w.shape
            cip, cop = cip or CI, cop or CO
            assert w.dtype == torch.float32 and w.is_contiguous()
            w_t = torch.empty(cop, KH * KW * cip, device=dev, dtype=dtype)
            w_d = torch.empty(cip, KH * KW * cop, device=dev, dtype=dtype)
            self.pairs.append((w_t, w_d))
            table += [w.data_ptr(), w_t.data_ptr(), w_d.data_ptr(), KH * KW, CI, CO, cip, cop]
            ntile = KH * KW * ((cip + chunk - 1) // chunk) * ((cop + chunk - 1) // chunk)
            chunks += [(t, i) for i in range(ntile)]
        self.table = torch.tensor(table, dtype=torch.int64).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int64).view(-1).to(dev)
        self.nchunks = len(chunks)
        # fp32 storage, three-term modes: the pre-split copies the forward (fp16 pieces of 2^8 w_t) and the data gradient (bf16 pieces of
        # w_d) would otherwise make per launch -- 120 launches of 5 us per ResNet-50 step -- written by ONE launch per refresh
        # (simclr_presplit_weights_multi) and handed to the library with SIMCLR_FMT_PS_W.  SIMCLR_PS_W=0: per-launch copies as before.
        import os
        self.ps = None
        if dtype == torch.float32 and os.environ.get('SIMCLR_PS_W', '1') not in ('', '0'):
            rows_f, rows_b, self.ps_t, self.ps_d = [], [], [], []
            for w_t, w_d in self.pairs:
                ok = w_t.shape[1] % 32 == 0 and w_d.shape[1] % 32 == 0
                pt = torch.empty_like(w_t) if ok else None
                pd = torch.empty_like(w_d) if ok else None
                self.ps_t.append(pt)
                self.ps_d.append(pd)
                if ok:
                    rows_f += [w_t.data_ptr(), pt.data_ptr(), w_t.numel() // 32, 1]
                    rows_b += [w_d.data_ptr(), pd.data_ptr(), w_d.numel() // 32, 0]
            if rows_f:
                self.ps = dict(f=torch.tensor(rows_f, dtype=torch.int64).to(dev), b=torch.tensor(rows_b, dtype=torch.int64).to(dev),
                               n=len(rows_f) // 4, maxb=max(rows_f[2::4] + rows_b[2::4]))

    def matches(self, entries, dtype):
        return dtype == self.dtype and self.key == tuple(w.data_ptr() for w, _, _ in entries)

    def run(self):
        lib().prep_weights_pair_multi(_p(self.table), _p(self.chunks), self.nchunks,
                                      DT_BF16 if self.dtype == torch.bfloat16 else DT_F32, _s())
        if self.ps is not None:
            fwd, bwd = _TERMS[0] == 13, _TERMS[1] == 3
            if fwd:
                lib().presplit_weights_multi(_p(self.ps['f']), self.ps['n'], self.ps['maxb'], _s())
            if bwd:
                lib().presplit_weights_multi(_p(self.ps['b']), self.ps['n'], self.ps['maxb'], _s())
            for (w_t, w_d), pt, pd in zip(self.pairs, self.ps_t, self.ps_d):
                w_t._psw = (pt, 13) if (fwd and pt is not None) else None
                w_d._psw = (pd, 3) if (bwd and pd is not None) else None
        return self.pairs


def conv2d_fwd(x, w_t, KH, KW, stride, pad, OH, OW, stats=None, out=None, store=True):
    """store=False (bf16, stats required): statistics-only pass, nothing is written and None is returned (first half
    of the fused conv + BatchNorm-apply forward, conv2d_fwd_bn_apply)."""
    V, IH, IW, Cin = x.shape
    Cout = w_t.shape[0]
    if not store:
        assert stats is not None and out is None
    elif out is None:
        out = torch.empty(V, OH, OW, Cout, device=x.device, dtype=x.dtype)
    M, K = V * OH * OW, KH * KW * Cin
    esz = x.element_size()
    wp, wf = _pw(w_t, True)
    _launch('conv_igemm_fwd', 2.0 * M * K * Cout, esz * (V * IH * IW * Cin + (M * Cout if store else 0) + K * Cout),
            lambda: lib().conv2d_fwd(_p(x), wp, _p(out), _p(stats), stats.shape[0] if stats is not None else 0, V,
                                     IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, dt(x) | _tf() | wf, _s()))
    return out


def bn_pivot_enabled(dtype):
    """Pivoted BatchNorm statistics of the fp32 convolutions (simclr_conv2d_fwd_pivoted); SIMCLR_BN_PIVOT=0: raw moments."""
    return dtype == torch.float32 and os.environ.get('SIMCLR_BN_PIVOT', '1') != '0'


def conv2d_fwd_with_stats(x, w_t, KH, KW, stride, pad, OH, OW, stats):
    """The forward convolution of a conv -> BatchNorm pair (tf2/resnet.py:183-208 followed by :31-78).  Returns (y, stats, sums):
    bf16 (and fp32 with SIMCLR_BN_PIVOT=0): the partial slots as conv2d_fwd fills them, sums None; fp32: the statistics are
    accumulated about a per-channel pivot and come back as this replica's raw fp64 moments `sums` [2, C] (stats None)."""
    if stats is None or not bn_pivot_enabled(x.dtype):
        return conv2d_fwd(x, w_t, KH, KW, stride, pad, OH, OW, stats=stats), stats, None
    V, IH, IW, Cin = x.shape
    Cout = w_t.shape[0]
    out = torch.empty(V, OH, OW, Cout, device=x.device, dtype=x.dtype)
    pivot = torch.empty(Cout, device=x.device, dtype=torch.float32)
    M, K = V * OH * OW, KH * KW * Cin
    wp, wf = _pw(w_t, True)
    _launch('conv_igemm_fwd', 2.0 * M * K * Cout, 4 * (V * IH * IW * Cin + M * Cout + K * Cout),
            lambda: lib().conv2d_fwd_pivoted(_p(x), wp, _p(out), _p(stats), stats.shape[0], _p(pivot), V, IH, IW, Cin, OH, OW,
                                             Cout, KH, KW, stride, pad, dt(x) | _tf() | wf, _s()))
    sums = torch.empty(2, Cout, device=x.device, dtype=torch.float64)
    lib().bn_reduce_slots_pivoted(_p(stats), stats.shape[0], Cout, _p(pivot), float(M), _p(sums), _s())
    return out, None, sums


def conv2d_fwd_bn_apply(x, w_t, KH, KW, stride, pad, OH, OW, scale, shift, res=None, relu=True, want_bits=False,
                        rscale=None, rshift=None):
    """y = act(T(conv(x)) * scale + shift + res) in ONE kernel (bf16; fp32 storage since round 6): what conv2d_fwd + bn_apply
    produce, bit for bit, without the convolution output travelling to memory.  Returns y or (y, relu_bits)."""
    V, IH, IW, Cin = x.shape
    Cout = w_t.shape[0]
    y = torch.empty(V, OH, OW, Cout, device=x.device, dtype=x.dtype)
    M, K = V * OH * OW, KH * KW * Cin
    esz = x.element_size()
    epc = 16 // esz                      # one mask byte per 16-byte chunk of y
    bits = torch.empty(M, Cout // epc, device=x.device, dtype=torch.uint8) if want_bits else None
    # algorithmic bytes = SURVEY 8(d)'s strict count (input + output + weights); the residual read and the mask write of
    # the fused epilogue (bn3's own traffic) are counted under impl_bytes only
    nb = esz * (V * IH * IW * Cin + M * Cout + K * Cout)
    wp, wf = _pw(w_t, True)
    _launch('conv_igemm_fwd', 2.0 * M * K * Cout, nb,
            impl_bytes=nb + (esz * M * Cout if res is not None else 0) + (M * Cout // epc if want_bits else 0),
            fn=lambda: lib().conv2d_fwd_bn_apply(_p(x), wp, _p(y), _p(scale), _p(shift), _p(res), _p(rscale), _p(rshift),
                                              int(relu), _p(bits), V,
                                              IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, dt(x) | _tf() | wf, _s()))
    return (y, bits) if want_bits else y


def sparse_dgrad_enabled(dtype):
    """fp32 storage: the data gradient of a stride-2 1x1 projection shortcut stores only the quarter of dx that receives a tap and the
    accumulating data gradient of the block's first convolution reads earlier data there only (simclr_conv2d_dgrad accumulate = 3 / 2):
    no zero fill of, and no read-back from, the other three quarters.  SIMCLR_SPARSE_DGRAD=0: zero fill + full accumulate."""
    return dtype == torch.float32 and os.environ.get('SIMCLR_SPARSE_DGRAD', '1') not in ('', '0')


def _acc_mode(out, accumulate):
    """accumulate argument of the data-gradient entry points: 2 when `out` came from a sparse store (tag `_sparse2`, cleared here)."""
    if not accumulate:
        assert not getattr(out, '_sparse2', False), 'a sparsely stored gradient must be completed by an accumulating data gradient'
        return 0
    if getattr(out, '_sparse2', False):
        out._sparse2 = False
        return 2
    return 1


def conv2d_dgrad(dy, w_d, KH, KW, stride, pad, IH, IW, out=None, accumulate=False, sparse=False):
    """sparse (fp32, stride-2 1x1): only the pixels with even row and column are written, the rest of the result is left
    UNINITIALISED and the tensor is tagged `_sparse2`: its only legal consumer is an accumulating conv2d_dgrad / conv2d_dgrad_bn."""
    V, OH, OW, Cout = dy.shape
    Cin = w_d.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty(V, IH, IW, Cin, device=dy.device, dtype=dy.dtype)
    if sparse:
        assert not accumulate and stride == 2 and KH == 1 and KW == 1 and pad == 0 and dy.dtype == torch.float32
    M, K = V * IH * IW, KH * KW * Cout
    esz = dy.element_size()
    fmt = FMT_PS_IN if ps_kind(dy) else 0
    assert ps_kind(dy) in (None, 'b16')
    wp, wf = _pw(w_d, False)
    acc = 3 if sparse else _acc_mode(out, accumulate)
    _launch('conv_igemm_dgrad', 2.0 * V * OH * OW * K * Cin, esz * (V * OH * OW * Cout + M * Cin + K * Cin),
            lambda: lib().conv2d_dgrad(_pp(dy), wp, _p(out), acc, V, IH, IW, Cin, OH, OW, Cout,
                                       KH, KW, stride, pad, dt(dy) | fmt | _tb() | wf, _s()))
    if sparse:
        out._sparse2 = True
    return out


def conv2d_dgrad_bn(dy, w_d, KH, KW, pad, IH, IW, bn, out=None, accumulate=False):
    """dgrad + fused BN-backward reduce of the producer BatchNormRelu.  `bn` = dict(x, mask, scale, shift,
    mean, rstd, mode).  Returns (dm, partial[NSLOT,2,Cin])."""
    V, OH, OW, Cout = dy.shape
    Cin = w_d.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty(V, IH, IW, Cin, device=dy.device, dtype=dy.dtype)
    partial = conv_stats(V * IH * IW, Cin, dy.device)
    K = KH * KW * Cout
    esz = dy.element_size()
    wp, wf = _pw(w_d, False)
    acc = _acc_mode(out, accumulate)
    _launch('conv_igemm_dgrad', 2.0 * V * OH * OW * K * Cin, esz * (V * OH * OW * Cout + V * IH * IW * Cin + K * Cin),
            impl_bytes=esz * (V * OH * OW * Cout + (1 + (bn['mode'] != 4) + (bn['mode'] == 1) + int(accumulate)) * V * IH * IW * Cin + K * Cin),
            fn=lambda: lib().conv2d_dgrad_bn(_pp(dy), wp, _p(out), acc, _p(bn.get('x')), _p(bn.get('mask')),
                                          _p(bn.get('scale')), _p(bn.get('shift')), _p(bn.get('mean')), _p(bn.get('rstd')),
                                          bn['mode'], _p(partial), partial.shape[0], V, IH, IW, Cin, OH, OW, Cout, KH, KW, 1,
                                          pad, dt(dy) | (FMT_PS_IN if ps_kind(dy) else 0) | _tb() | wf, _s()))
    return out, partial


_wgrad_ws = {}


def _workspace(nbytes, device, key='ws'):
    """One scratch buffer per (purpose, device, STREAM): launches on one stream are ordered, so they can share it; a launch
    on the weight-gradient side stream (SIMCLR_WGRAD_STREAM=1) gets its own, so it can never race a main-stream user of the
    same scratch space.  A buffer that has to grow is replaced, the old one stays referenced by the launches
    already queued on its stream (same stream: the allocator orders the reuse)."""
    import os
    # (default: one stream, one buffer -- the configuration the GPU suite runs; the per-stream key only exists with the side stream)
    side = os.environ.get('SIMCLR_WGRAD_STREAM', '0') not in ('', '0')
    k = (key, device, torch.cuda.current_stream(device).cuda_stream) if side else (key, device)
    buf = _wgrad_ws.get(k)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _wgrad_ws[k] = buf
    return buf


def conv2d_wgrad(x, dy, KH, KW, stride, pad, Cin=None, pixpitch=None, out=None, accumulate=False):
    V, IH, IW = x.shape[0], x.shape[1], x.shape[2]
    if Cin is None:
        Cin = x.shape[3]
    if pixpitch is None:
        pixpitch = x.shape[3]
    _, OH, OW, Cout = dy.shape
    if out is None:
        assert not accumulate
        out = torch.empty(KH * KW * Cin, Cout, device=x.device, dtype=torch.float32)
    nbytes = lib().conv2d_wgrad_workspace_bytes(V, OH, OW, Cin, Cout, KH, KW, dt(x))
    ws = _workspace(nbytes, x.device)
    M, K = V * OH * OW, KH * KW * Cin
    esz = x.element_size()
    _launch('conv_wgrad', 2.0 * M * K * Cout, esz * (V * IH * IW * pixpitch + M * Cout) + 4 * K * Cout,
            lambda: lib().conv2d_wgrad(_p(x), _pp(dy), _p(out), int(accumulate), _p(ws), V, IH, IW, Cin, pixpitch,
                                       OH, OW, Cout, KH, KW, stride, pad, dt(x) | (FMT_PS_IN if ps_kind(dy) else 0) | _tb(), _s()))
    return out


def stem_geometry(H, W, KH, KW, stride):
    """Packed-input geometry for the stem conv (Conv2dFixedPadding: pad (k-1)//2 before)."""
    pad = (KH - 1) // 2
    OH = (H + (KH - 1) - KH) // stride + 1
    OW = (W + (KW - 1) - KW) // stride + 1
    KWP = 8
    KHP = KH
    HP = (OH - 1) * stride + KHP
    WP = (OW - 1) * stride + KWP
    HP = max(HP, H + pad)
    WP = max(WP, W + pad)
    WP = (WP + 1) // 2 * 2     # 16-byte aligned rows for bf16 (8 B / pixel)
    return dict(pad=pad, OH=OH, OW=OW, KHP=KHP, KWP=KWP, HP=HP, WP=WP)


def pack_views(images, k, geo, dtype, with_presplit=False):
    """with_presplit (fp32): also returns presplit_packed(xp), written in the same pass -> (xp, xq)."""
    b, H, W, C = images.shape
    assert C == 3 * k and images.dtype == torch.float32
    xp = torch.empty(k * b, geo['HP'], geo['WP'], 4, device=images.device, dtype=dtype)
    if with_presplit:
        assert dtype == torch.float32
        xq = torch.empty_like(xp)
        lib().pack_views_ps(_p(images), _p(xp), _p(xq), b, H, W, k, geo['HP'], geo['WP'], geo['pad'], _s())
        return xp, xq
    lib().pack_views(_p(images), _p(xp), b, H, W, k, geo['HP'], geo['WP'], geo['pad'], dt(xp), _s())
    return xp


def stem_conv_fwd(xp, w_s, geo, stride, stats=None):
    V = xp.shape[0]
    Cout = w_s.shape[0]
    y = torch.empty(V, geo['OH'], geo['OW'], Cout, device=xp.device, dtype=xp.dtype)
    M = V * geo['OH'] * geo['OW']
    _launch('stem_conv_fwd', 2.0 * M * 147 * Cout, xp.element_size() * (xp.numel() + M * Cout),
            lambda: lib().stem_conv_fwd(_p(xp), _p(w_s), _p(y), _p(stats), stats.shape[0] if stats is not None else 0, V,
                                        geo['HP'], geo['WP'], geo['OH'], geo['OW'], Cout, geo['KHP'], geo['KWP'],
                                        stride, dt(xp) | _tf(), _s()))
    return y


def stem_wgrad_ps_supported(geo, KH, stride, Cout):
    """Does the pre-split stem weight gradient (simclr_stem_wgrad_ps) cover this stem?  fp32 storage with three bf16 backward terms
    (ps_backward_enabled), the 7x7 / stride-2 stem with 64 output channels; SIMCLR_STEM_WGRAD_PS=0 keeps the multi-tap kernel."""
    import os
    if os.environ.get('SIMCLR_STEM_WGRAD_PS', '1') in ('', '0') or not ps_backward_enabled():
        return False
    return bool(lib().stem_wgrad_ps_supported(KH, geo['KWP'], stride, Cout))


def presplit_packed(xp):
    """The packed views [V, HP, WP, 4] fp32 as (four bf16 hi pieces | four bf16 lo pieces) per pixel: the image operand of
    stem_conv_wgrad when the gradient arrives pre-split."""
    assert xp.dtype == torch.float32 and xp.shape[-1] == 4 and xp.is_contiguous()
    xq = torch.empty_like(xp)
    lib().presplit_packed(_p(xp), _p(xq), xp.numel() // 4, _s())
    return xq


def stem_conv_wgrad(xp, dy, geo, KH, KW, stride, out=None, accumulate=False, xq=None):
    """dW (HWIO [KH,KW,3,Cout] fp32) of the stem from the packed input.  A pre-split dy (bn_bwd_apply(ps_out=True), tagged `_ps`) takes
    simclr_stem_wgrad_ps with xq = presplit_packed(xp) (computed here when the caller has not kept one)."""
    Cout = dy.shape[3]
    kp = geo['KHP'] * geo['KWP'] * 4
    tmp = torch.empty(kp, Cout, device=dy.device, dtype=torch.float32)
    if ps_kind(dy):
        assert ps_kind(dy) == 'b16' and stem_wgrad_ps_supported(geo, KH, stride, Cout), 'pre-split dy without a pre-split stem kernel'
        if xq is None:
            xq = presplit_packed(xp)
        V, OH, OW = dy.shape[0], dy.shape[1], dy.shape[2]
        ws = _workspace(lib().stem_wgrad_ps_workspace_bytes(V, OH, OW, geo['KHP']), dy.device)
        M = V * OH * OW
        _launch('conv_wgrad', 2.0 * M * kp * Cout, 4 * (xp.numel() + M * Cout) + 4 * kp * Cout,
                lambda: lib().stem_wgrad_ps(_p(xq), _pp(dy), _p(tmp), 0, _p(ws), V, geo['HP'], geo['WP'], OH, OW, Cout, geo['KHP'],
                                            geo['KWP'], stride, _s()))
        if out is None:
            out = torch.empty(KH, KW, 3, Cout, device=dy.device, dtype=torch.float32)
        lib().unpack_stem_dw(_p(tmp), _p(out), KH, KW, 3, Cout, geo['KWP'], int(accumulate), _s())
        return out
    # one 'tap' per kernel row: KWP*4 = 32 contiguous elements, pixel pitch 4
    conv2d_wgrad(xp, dy, geo['KHP'], 1, stride, 0, Cin=geo['KWP'] * 4, pixpitch=4, out=tmp)
    if out is None:
        out = torch.empty(KH, KW, 3, Cout, device=dy.device, dtype=torch.float32)
    lib().unpack_stem_dw(_p(tmp), _p(out), KH, KW, 3, Cout, geo['KWP'], int(accumulate), _s())
    return out


# ---------------------------------------------------------------- batch norm
class _StatsArena:
    """All per-layer statistic partials of one step live in one buffer that is zeroed by a single
    memset at the start of the step (instead of ~220 tiny fill kernels)."""

    def __init__(self):
        self.buf = None
        self.ptr = 0
        self.high = 0
        self.missed = 0
        self.want = 0

    def begin_step(self, device, nfloats=64 << 20):
        want = max(nfloats, getattr(self, 'want', 0))
        if self.buf is None or self.buf.device != torch.device(device) or self.buf.numel() < want:
            self.buf = torch.zeros(want, device=device, dtype=torch.float32)
        else:
            self.buf[:max(self.high, 1)].zero_()       # only what the previous step handed out
        self.ptr = 0
        self.high = 0
        self.missed = 0

    def take(self, n, device):
        n_al = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != torch.device(device) or self.ptr + n_al > self.buf.numel():
            if self.buf is not None and self.ptr < (1 << 60):
                self.missed += n_al                        # arena too small: grow it for the next step
                self.want = self.buf.numel() + 2 * self.missed
            return None
        out = self.buf[self.ptr:self.ptr + n]
        self.ptr += n_al
        self.high = max(self.high, self.ptr)
        return out


_ARENA = _StatsArena()


# (forward terms, backward terms) of simclr_set_f32_matmul; 13 = three split-FP16 terms (11-bit pieces: the accuracy of six bf16
# terms at half the MFMA work; forward only -- gradients span too many binades for fp16 pieces)
F32_MATMUL_TERMS = {'exact': (0, 0), 'bf16x3': (3, 3), 'bf16x6': (6, 6), 'bf16x6_3': (6, 3), 'f16x3_3': (13, 3)}


_TERMS = [0, 0]        # (forward, backward) terms every convolution / dense call of this module passes to the library


def _tf():
    """dtype field of a FORWARD call: this call's matrix arithmetic, SIMCLR_FMT_TERMS(t) = (t + 1) << 12 (csrc/conv.hip terms_of).
    The library's process-wide default (simclr_set_f32_matmul) is never consulted by calls that carry the field: the C ABI is
    re-entrant across threads / streams with different modes."""
    return (_TERMS[0] + 1) << 12


def _tb():
    return (_TERMS[1] + 1) << 12


def set_f32_matmul(mode):
    """Matrix arithmetic of the fp32 convolution / dense kernels (FLAGS.f32_matmul).  The mode is module state of THIS Python layer, sent
    with every call (_tf / _tb); the library's own process-wide default is set too, for callers of the C ABI that pass no field."""
    if mode not in F32_MATMUL_TERMS:
        raise ValueError('f32_matmul must be one of %s, got %r' % (sorted(F32_MATMUL_TERMS), mode))
    fwd, bwd = F32_MATMUL_TERMS[mode]
    _TERMS[0], _TERMS[1] = fwd, bwd
    L = lib()
    if (L.get_f32_matmul(0), L.get_f32_matmul(1)) != (fwd, bwd):
        L.set_f32_matmul(fwd, bwd)


def select_f32_matmul(inference=False):
    """Apply FLAGS.f32_matmul -- for compute_dtype='f32' ONLY: with a bf16 encoder and head_dtype='f32' the fp32
    projection / linear-eval heads exist to keep the loss gradient exact, so they always run the exact fp32-input MFMA.
    inference: a forward through the moving BatchNorm statistics -- 'f16x3_3' falls back to 'bf16x6_3' (fp16 pieces need O(1) operands,
    which only batch statistics guarantee)."""
    from .flags import FLAGS
    mode = getattr(FLAGS, 'f32_matmul', 'exact') if getattr(FLAGS, 'compute_dtype', 'f32') == 'f32' else 'exact'
    if inference and mode == 'f16x3_3':
        mode = 'bf16x6_3'
    set_f32_matmul(mode)


def begin_step(device):
    """Zero the statistics arena and select the fp32 matrix arithmetic; call once at the start of every training step."""
    select_f32_matmul()
    _ARENA.begin_step(device)


def check_split_tail_health():
    """Raise if a workgroup sharing a left-over tile of the persistent convolution grid ever timed out waiting for a partner
    (simclr_conv2d_split_tail_timeouts: sticky device counters; synchronous -- call where the host synchronises anyway)."""
    n = ctypes.c_uint(0)
    lib().conv2d_split_tail_timeouts(ctypes.byref(n))
    if n.value:
        raise RuntimeError('%d split-tail partner(s) of the persistent convolution grid timed out: results of those tiles are wrong '
                           '(SIMCLR_IGEMM_SPLIT=0 disables the split tail)' % n.value)
    return 0


def end_step():
    """Stop handing out arena slices (calls outside a step fall back to fresh zero tensors)."""
    _ARENA.ptr = 1 << 62


def new_stats(C, device, slots=NSLOT):
    """Zeroed partial-statistics buffer [slots, 2, C] (from the per-step arena when inside a step)."""
    t = _ARENA.take(slots * 2 * C, device)
    if t is None:
        return torch.zeros(slots, 2, C, device=device, dtype=torch.float32)
    return t.view(slots, 2, C)


def conv_stats(M, C, device):
    """Statistics buffer for conv2d_fwd / conv2d_dgrad_bn over an [M, C] output with one slot per producing workgroup
    (simclr_conv2d_stats_slots): plain stores instead of float atomics -> run-to-run deterministic BatchNorm."""
    return new_stats(C, device, lib().conv2d_stats_slots(M, C))


def stem_stats(M, C, device):
    return new_stats(C, device, lib().stem_stats_slots(M))


def bn_reduce_slots(partial):
    C = partial.shape[2]
    sums = torch.empty(2, C, device=partial.device, dtype=torch.float64)
    lib().bn_reduce_slots(_p(partial), partial.shape[0], C, _p(sums), _s())
    return sums


def bn_finalize(sums, count, gamma, beta, moving_mean, moving_var, decay, eps=1e-5, partial=None):
    """sums [2,C] fp64 (cross-replica path) or partial [NSLOT,2,C] fp32 (single replica: fused reduce)."""
    src = partial if partial is not None else sums
    C = src.shape[-1]
    dev = src.device
    mean, rstd, scale, shift = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(4))
    lib().bn_finalize(_p(sums) if partial is None else None, _p(partial), partial.shape[0] if partial is not None else 0,
                      float(count), C, _p(gamma), _p(beta), _p(moving_mean), _p(moving_var),
                      float(decay), float(eps), _p(mean), _p(rstd), _p(scale), _p(shift), _s())
    return mean, rstd, scale, shift


def bn_apply(x, scale, shift, relu, res=None, rscale=None, rshift=None, out=None, want_bits=False):
    """want_bits: also return the ReLU mask (out > 0) as uint8 [rows, C/epc], one byte per 16-byte chunk
    (the operand of conv2d_dgrad_bn mode 3)."""
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    bits = torch.empty(rows, C // (16 // x.element_size()), device=x.device, dtype=torch.uint8) if want_bits else None
    lib().bn_apply(_p(x), _p(scale), _p(shift), _p(res), _p(rscale), _p(rshift), _p(out), _p(bits), rows, C,
                   int(relu), dt(x), _s())
    return (out, bits) if want_bits else out


def bn_bwd_reduce(dy, x, mask_src, scale, shift, mean, rstd, mask_mode):
    C = x.shape[-1]
    rows = x.numel() // C
    partial = new_stats(C, x.device, lib().bn_bwd_reduce_slots(rows, C, dt(x)))
    lib().bn_bwd_reduce(_p(dy), _p(x), _p(mask_src), _p(scale), _p(shift), _p(mean), _p(rstd), rows, C,
                        mask_mode, _p(partial), partial.shape[0], dt(x), _s())
    return partial


def bn_bwd_finalize(local_sums, global_sums, count, dgamma, dbeta, accumulate=False, partial=None):
    src = partial if partial is not None else local_sums
    C = src.shape[-1]
    c1 = torch.empty(C, device=src.device, dtype=torch.float32)
    c2 = torch.empty(C, device=src.device, dtype=torch.float32)
    lib().bn_bwd_finalize(_p(local_sums) if partial is None else None, _p(global_sums) if partial is None else None,
                          _p(partial), partial.shape[0] if partial is not None else 0, float(count), C,
                          _p(dgamma), _p(dbeta), int(accumulate), _p(c1), _p(c2), _s())
    return c1, c2


def bn_bwd_apply(dy, x, mask_src, scale, shift, mean, rstd, c1, c2, mask_mode, want_masked=False, out=None, ps_out=False):
    """ps_out (fp32, C % 32 == 0): dx in the pre-split block format with bf16 pieces (tagged `_ps`), for the data-gradient and
    weight-gradient GEMMs of the convolution in front of this BatchNorm."""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x) if out is None else out
    dmasked = torch.empty_like(x) if want_masked else None
    ps_out = bool(ps_out) and x.dtype == torch.float32 and C % 32 == 0
    lib().bn_bwd_apply(_p(dy), _p(x), _p(mask_src), _p(scale), _p(shift), _p(mean), _p(rstd), _p(c1), _p(c2),
                       rows, C, mask_mode, _pp(dx), _p(dmasked), dt(x) | (FMT_PS_OUT if ps_out else 0), _s())
    if ps_out:
        dx._ps = 'b16'
    return dx, dmasked


# ---------------------------------------------------------------- pooling
def same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def bnrelu_maxpool_fwd(x, scale, shift, ksz=3, stride=2):
    V, H, W, C = x.shape
    OH, pt = same_pad(H, ksz, stride)
    OW, pl = same_pad(W, ksz, stride)
    y = torch.empty(V, OH, OW, C, device=x.device, dtype=x.dtype)
    arg = torch.empty(V, OH, OW, C, device=x.device, dtype=torch.uint8)
    lib().bnrelu_maxpool_fwd(_p(x), _p(scale), _p(shift), _p(y), _p(arg), V, H, W, C, OH, OW, ksz, stride,
                             pt, pl, dt(x), _s())
    return y, arg


def maxpool_bwd(dy, arg, H, W, ksz=3, stride=2):
    V, OH, OW, C = dy.shape
    _, pt = same_pad(H, ksz, stride)
    _, pl = same_pad(W, ksz, stride)
    dx = torch.empty(V, H, W, C, device=dy.device, dtype=dy.dtype)
    lib().maxpool_bwd(_p(dy), _p(arg), _p(dx), V, H, W, C, OH, OW, ksz, stride, pt, pl, dt(dy), _s())
    return dx


def global_avgpool_fwd(x, out_dtype=None):
    """reduce_mean over H, W (tf2/resnet.py:693-696).  out_dtype=torch.float32: fp32 means from bf16 activations."""
    V, H, W, C = x.shape
    if out_dtype is None or out_dtype == x.dtype:
        y = torch.empty(V, C, device=x.device, dtype=x.dtype)
        lib().global_avgpool_fwd(_p(x), _p(y), V, H * W, C, dt(x), _s())
        return y
    assert out_dtype == torch.float32
    y = torch.empty(V, C, device=x.device, dtype=torch.float32)
    lib().global_avgpool_fwd_f32(_p(x), _p(y), V, H * W, C, dt(x), _s())
    return y


def global_avgpool_bwd(dy, H, W, mask_src=None):
    V, C = dy.shape
    dx = torch.empty(V, H, W, C, device=dy.device, dtype=dy.dtype)
    lib().global_avgpool_bwd(_p(dy), _p(mask_src), _p(dx), V, H * W, C, dt(dy), _s())
    return dx


def batch_blur(images, filt, selector):
    """images f32 [b,H,W,3k]; filt [k,K]; selector [k,b] of 0/1."""
    b, H, W, C = images.shape
    k, K = filt.shape
    assert C == 3 * k and images.dtype == torch.float32
    tmp = torch.empty_like(images)
    out = torch.empty_like(images)
    lib().batch_blur(_p(images), _p(tmp), _p(out), _p(filt), _p(selector), b, H, W, k, K, _s())
    return out


def avgpool2_fwd(x, stride):
    V, H, W, C = x.shape
    OH, OW = (H, W) if stride == 1 else ((H + 1) // 2, (W + 1) // 2)
    y = torch.empty(V, OH, OW, C, device=x.device, dtype=x.dtype)
    lib().avgpool2_fwd(_p(x), _p(y), V, H, W, C, stride, dt(x), _s())
    return y


def avgpool2_bwd(dy, H, W, stride):
    V, OH, OW, C = dy.shape
    dx = torch.empty(V, H, W, C, device=dy.device, dtype=dy.dtype)
    lib().avgpool2_bwd(_p(dy), _p(dx), V, H, W, C, stride, dt(dy), _s())
    return dx


# ---------------------------------------------------------------- selective kernel unit
def sk_pool_fwd(a, f, gpitch):
    V, H, W, _ = a.shape
    g = torch.zeros(V, gpitch, device=a.device, dtype=a.dtype)
    lib().sk_pool_fwd(_p(a), _p(g), V, H * W, f, gpitch, dt(a), _s())
    return g


def sk_mix_fwd(a, l, f):
    V, H, W, _ = a.shape
    out = torch.empty(V, H, W, f, device=a.device, dtype=a.dtype)
    lib().sk_mix_fwd(_p(a), _p(l), _p(out), V, H * W, f, l.shape[1], dt(a), _s())
    return out


def sk_mix_bwd_logits(a, l, dout, f):
    V, H, W, _ = a.shape
    dl = torch.zeros_like(l)
    lib().sk_mix_bwd_logits(_p(a), _p(l), _p(dout), _p(dl), V, H * W, f, l.shape[1], dt(a), _s())
    return dl


def sk_mix_bwd_streams(l, dout, dg, f):
    V, H, W, _ = dout.shape
    da = torch.empty(V, H, W, 2 * f, device=dout.device, dtype=dout.dtype)
    lib().sk_mix_bwd_streams(_p(l), _p(dout), _p(dg), _p(da), V, H * W, f, l.shape[1], dg.shape[1], dt(dout), _s())
    return da


# ---------------------------------------------------------------- supervised head / misc
def bias_softmax_xent(z, bias, labels, nclass, gscale, out):
    rows, cpad = z.shape
    dlogits = torch.empty_like(z)
    lib().bias_softmax_xent(_p(z), _p(bias), _p(labels), rows, labels.shape[0], nclass, cpad, float(gscale),
                            _p(dlogits), _p(out), dt(z), _s())
    return dlogits


def colsum(x, cvalid, out, accumulate=False):
    rows, C = x.shape
    lib().colsum(_p(x), rows, C, cvalid, _p(out), int(accumulate), dt(x), _s())
    return out


def cast(x, dtype, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    lib().cast(_p(x), _p(out), x.numel(), dt(x), dt(out), _s())
    return out


def axpy_f32(a, x, y):
    lib().axpy_f32(float(a), _p(x), _p(y), x.numel(), _s())


def step_scalars(n, device):
    """n zeroed fp32 scalars that live until the next begin_step (a slice of the per-step arena: no fill launch); outside
    a step a fresh zero tensor."""
    t = _ARENA.take(n, device)
    return t if t is not None else torch.zeros(n, device=device, dtype=torch.float32)


def accumulate_scalars(srcs, dst=None, scales=None, total=None, total_mask=0, copy=None):
    """dst[i] += scales[i] * srcs[i][0] for up to 16 device scalars in ONE launch; total[0] = sum of the scaled terms
    selected by total_mask; copy[i] = the scaled term (simclr_accumulate_scalars)."""
    assert 1 <= len(srcs) <= 16, 'accumulate_scalars: 1..16 scalars per launch, got %d' % len(srcs)
    import ctypes
    n = len(srcs)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    sc = (ctypes.c_float * n)(*[float(x) for x in scales]) if scales is not None else None
    lib().accumulate_scalars(ptrs, sc, n, _p(dst), _p(total), int(total_mask), _p(copy), _s())


def l2_loss_f32(x, out):
    lib().l2_loss_f32(_p(x), x.numel(), _p(out), _s())


# ---------------------------------------------------------------- two-view augmentation
def augment_views(src, params, H, W, out=None):
    """src [b, Hs, Ws, 3] float32 in [0,1] or uint8; params [b, views, 16] float32 (simclr_amd.data_util.PARAM_FIELDS).
    Returns [b, H, W, 3*views] float32 in [0,1]."""
    b, Hs, Ws, C = src.shape
    assert C == 3 and params.dtype == torch.float32 and params.shape[0] == b and params.shape[2] == 16
    views = params.shape[1]
    if src.dtype == torch.uint8:
        code = 2
    elif src.dtype == torch.float32:
        code = DT_F32
    else:
        raise TypeError('augment_views: source images must be uint8 or float32, got %s' % src.dtype)
    if out is None:
        out = torch.empty(b, H, W, 3 * views, device=src.device, dtype=torch.float32)
    ws = _workspace(lib().augment_workspace_bytes(b, views, H, W), src.device, key='augment')
    lib().augment_views(_p(src), code, _p(params), _p(ws), _p(out), b, views, Hs, Ws, H, W, _s())
    return out


# ---------------------------------------------------------------- BatchNorm backward folded into the producing 1x1 conv
def bn_fold_coeffs(scale, mean, rstd, c1, c2):
    C = scale.shape[0]
    a, b, d = (torch.empty(C, device=scale.device, dtype=torch.float32) for _ in range(3))
    lib().bn_fold_coeffs(_p(scale), _p(mean), _p(rstd), _p(c1), _p(c2), _p(a), _p(b), _p(d), C, _s())
    return a, b, d


def bn_fold_pre(w_d, scale, mean, rstd, c1, c2):
    """w_d [K][N] (compute copy, = the conv's dgrad layout) + the BN-backward quantities.  Returns (a, b, d [N],
    wb [K,N] f32 = w*b, wext [K, N+K] (first N columns = w*a, the rest filled by bn_fold_post), e [K] f32 = w d)."""
    K, N = w_d.shape
    dev = w_d.device
    a, b, d = (torch.empty(N, device=dev, dtype=torch.float32) for _ in range(3))
    wb = torch.empty(K, N, device=dev, dtype=torch.float32)
    wext = torch.empty(K, N + K, device=dev, dtype=w_d.dtype)
    e = torch.empty(K, device=dev, dtype=torch.float32)
    lib().bn_fold_pre(_p(w_d), _p(scale), _p(mean), _p(rstd), _p(c1), _p(c2), _p(a), _p(b), _p(d), _p(wb), _p(wext), _p(e),
                      K, N, dt(w_d), _s())
    return a, b, d, wb, wext, e


def bn_fold_post(t1, gw, cs, a, b, d, q, dw, wext, accumulate=False):
    """cs: column sums of h, fp64 [>=K] (bn_reduce_slots output) or fp32 [K] (conv2d_gram)."""
    K, N = t1.shape
    assert tuple(q.shape) == (K, K) and tuple(dw.shape) == (K, N) and cs.is_contiguous()
    c64, c32 = (_p(cs), None) if cs.dtype == torch.float64 else (None, _p(cs))
    lib().bn_fold_post(_p(t1), _p(gw), c64, c32, _p(a), _p(b), _p(d), _p(q), _p(dw), _p(wext), K, N, int(accumulate),
                       dt(wext), _s())
    return dw


def conv2d_gram(h):
    """h [..., K] activation -> (h^T h [K, K] fp32, colsum(h) [K] fp32), streaming h once (K in {64, 128, 256 bf16})."""
    K = h.shape[-1]
    M = h.numel() // K
    out = torch.empty(K * K + K, device=h.device, dtype=torch.float32)
    ws = _workspace(lib().conv2d_gram_workspace_bytes(M, K, dt(h)), h.device)
    esz = h.element_size()
    _launch('conv_wgrad', 2.0 * M * K * K, esz * M * K + 4 * K * K,
            lambda: lib().conv2d_gram(_p(h), _p(out), _p(ws), M, K, dt(h) | (_tf() if h.dtype == torch.float32 else 0), _s()))
    return out[:K * K].view(K, K), out[K * K:]


def bn_sums_from_gram(gw, w_kn, cs):
    """(sum c, sum c^2) per output channel of c = h W from GW = (h^T h) W [K, N], W [K, N] fp32 and colsum(h) -> fp64 [2, N]."""
    K, N = gw.shape
    assert tuple(w_kn.shape) == (K, N) and gw.dtype == torch.float32 and w_kn.dtype == torch.float32
    sums = torch.empty(2, N, device=gw.device, dtype=torch.float64)
    c64, c32 = (_p(cs), None) if cs.dtype == torch.float64 else (None, _p(cs))
    lib().bn_sums_from_gram(_p(gw), _p(w_kn), c64, c32, K, N, _p(sums), _s())
    return sums


def gram_supported(K, dtype):
    return K in (64, 128) or (K == 256 and dtype == torch.bfloat16)


def small_gemm_nt(A, B):
    """A [M, K] fp32, B [N, K] fp32 -> A B^T [M, N] fp32 (exact f32 MFMA; small matrices)."""
    M, K = A.shape
    N = B.shape[0]
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and B.shape[1] == K
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    lib().small_gemm_nt_f32(_p(A), _p(B), _p(C), M, N, K, _s())
    return C


def conv2d_dgrad_bn_ext(dm, h, wext, bias, bn, out=None, accumulate=False):
    """1x1 stride-1 dgrad reading (dm [V,H,W,N], h [V,H,W,K]) with the K-extended weights of bn_fold_pre/post and the bias
    W d, plus the fused BN-backward reduce of the producer BN of h (`bn` as in conv2d_dgrad_bn).  Returns (dm_in, partial)."""
    assert not getattr(out, '_sparse2', False), 'a sparsely stored gradient needs conv2d_dgrad / conv2d_dgrad_bn to complete it'
    V, H, W, N = dm.shape
    K = h.shape[3]
    if out is None:
        assert not accumulate
        out = torch.empty(V, H, W, K, device=dm.device, dtype=dm.dtype)
    partial = conv_stats(V * H * W, K, dm.device)
    esz = dm.element_size()
    M = V * H * W
    _launch('conv_igemm_dgrad', 2.0 * M * N * K, esz * (M * N + M * K + N * K),
            impl_bytes=esz * (M * N + (3 + (bn['mode'] == 1) + int(accumulate)) * M * K + (N + K) * K),
            fn=lambda: lib().conv2d_dgrad_bn_ext(_p(dm), _p(h), _p(wext), _p(bias), _p(out), int(accumulate), _p(bn['x']),
                                                 _p(bn.get('mask')), _p(bn.get('scale')), _p(bn.get('shift')), _p(bn['mean']),
                                                 _p(bn['rstd']), bn['mode'], _p(partial), partial.shape[0], V, H, W, K, N,
                                                 dt(dm) | _tb(), _s()))
    return out, partial


# ---------------------------------------------------------------- stem backward fused with the max-pool backward
def bn_bwd_reduce_pool(dy, arg, x, scale, shift, mean, rstd, ksz=3, stride=2):
    V, H, W, C = x.shape
    _, OH, OW, _ = dy.shape
    _, pt = same_pad(H, ksz, stride)
    _, pl = same_pad(W, ksz, stride)
    partial = new_stats(C, x.device, lib().bn_bwd_pool_slots(V * H * W, C, dt(x)))
    lib().bn_bwd_reduce_pool(_p(dy), _p(arg), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), V, H, W, C, OH, OW, ksz,
                             stride, pt, pl, _p(partial), partial.shape[0], dt(x), _s())
    return partial


def bn_bwd_apply_pool(dy, arg, x, scale, shift, mean, rstd, c1, c2, ksz=3, stride=2, ps_out=False):
    """ps_out (fp32, C % 32 == 0): dx in the pre-split block format with bf16 pieces (tagged `_ps`), for the stem's weight gradient."""
    V, H, W, C = x.shape
    _, OH, OW, _ = dy.shape
    _, pt = same_pad(H, ksz, stride)
    _, pl = same_pad(W, ksz, stride)
    dx = torch.empty_like(x)
    lib().bn_bwd_apply_pool(_p(dy), _p(arg), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(c1), _p(c2), _p(dx), V, H,
                            W, C, OH, OW, ksz, stride, pt, pl, dt(x) | (FMT_PS_OUT if ps_out else 0), _s())
    if ps_out:
        dx._ps = 'b16'
    return dx


def bn_fold_s2(t1, w_d, mean, rstd, sums):
    """sums [2, N] fp64 with sums[0] = sum(dm): fills sums[1] = sum(dm * x^) of the BatchNorm behind c = h W from
    t1 = h^T dm [K, N] and the weight copy w_d [K, N] -- no pass over c."""
    K, N = t1.shape
    assert sums.dtype == torch.float64 and tuple(sums.shape) == (2, N) and tuple(w_d.shape) == (K, N)
    lib().bn_fold_s2(_p(t1), _p(w_d), _p(mean), _p(rstd), _p(sums), K, N, dt(w_d), _s())
    return sums


def conv2d_dgrad_ext(dm, h, wext, bias, out=None, accumulate=False):
    """K-extended 1x1 dgrad with a plain epilogue (see conv2d_dgrad_bn_ext): dx [+]= dm (a*W)^T + h Q + W d."""
    assert not getattr(out, '_sparse2', False), 'a sparsely stored gradient needs conv2d_dgrad / conv2d_dgrad_bn to complete it'
    V, H, W, N = dm.shape
    K = h.shape[3]
    if out is None:
        assert not accumulate
        out = torch.empty(V, H, W, K, device=dm.device, dtype=dm.dtype)
    esz = dm.element_size()
    M = V * H * W
    _launch('conv_igemm_dgrad', 2.0 * M * N * K, esz * (M * N + M * K + N * K),
            impl_bytes=esz * (M * N + (2 + int(accumulate)) * M * K + (N + K) * K),
            fn=lambda: lib().conv2d_dgrad_ext(_p(dm), _p(h), _p(wext), _p(bias), _p(out), int(accumulate), V, H, W, K, N,
                                              dt(dm) | _tb(), _s()))
    return out
