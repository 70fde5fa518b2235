"""ResNet encoder -- drop-in mirror of /root/reference/tf2/resnet.py on the HIP kernels.

Same public surface as the reference module: `resnet(resnet_depth, width_multiplier,
cifar_stem, data_format, dropblock_keep_probs, dropblock_size)` returns a `Resnet` whose
`__call__(inputs, training)` maps NHWC images to [B, 2048*w] (512*w for depth 18/34);
classes `BatchNormRelu`, `Conv2dFixedPadding`, `ResidualBlock`, `BottleneckBlock`,
`BlockGroup` keep the reference's constructor arguments.  There is no autodiff here:
every layer also has a hand-written `backward` (what `tape.gradient`, tf2/run.py:621,
derives), driven in reverse by `Resnet.backward`.

MI355X-first differences (numerically equivalent to the reference graph):
  * conv epilogues emit the BatchNorm statistics, so BN never re-reads its input for them;
  * the block tail `relu(bn(x) + shortcut)` (resnet.py:382,487) and the projection
    shortcut's BN apply are one fused elementwise pass;
  * stem BN + ReLU + max-pool (resnet.py:602-611) are one pass over the stem output;
  * the 3-channel input is packed once (pad + view split, tf2/model.py:250-259) so the
    stem conv needs no bounds checks.
Selective kernels (sk_ratio>0: SK_Conv2D resnet.py:217-277, ResNet-D stem :566-591 and avg-pool
shortcut :330-338/:400-408) are built; channel counts that are not a multiple of 64 (SK squeeze
dim, 32*w stem) are zero-padded internally.  Not built (fail loudly): SE (se_ratio>0, :280-311),
DropBlock (:81-157, unreachable in the reference too), channels_first.
"""
import math
import weakref

import torch

from . import ops
from .comm import collectives_on, num_replicas
from .flags import FLAGS
from .lars_optimizer import Variable

BATCH_NORM_EPSILON = 1e-5  # tf2/resnet.py:28


# --------------------------------------------------------------------------- runtime context
class _Runtime:
    """Process-wide build context (the stand-in for Keras name scopes + tf.distribute scope)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.counters = {}
        self.scope = []
        self.strategy = None
        self.device = 'cuda'
        self.seed = 0
        self.weights_version = 0     # bumped by the optimizer step; compute copies refresh lazily
        self._wgrad_stream = None
        self._consts = {}
        self.convs = []              # every built Conv2dFixedPadding with a (w_t, w_d) pair: refreshed in one launch
        self._conv_batch = None

    def const(self, C, value):
        """Cached constant fp32 vector on the device (means 0 / rstd 1 for plain column sums)."""
        key = (C, float(value), str(self.device))
        t = self._consts.get(key)
        if t is None:
            t = torch.full((C,), float(value), device=self.device, dtype=torch.float32)
            self._consts[key] = t
        return t

    def refresh_conv_weights(self):
        """Rewrite the compute copies (w_t, w_d) of EVERY registered convolution from the fp32 masters with one launch
        and mark them current.  Called by the first stale layer after an optimizer step / checkpoint restore."""
        self.convs = [r for r in self.convs if r() is not None]      # weak references: models may have been dropped
        convs = [r() for r in self.convs]
        convs = [c for c in convs if c is not None and c.kernel is not None]
        entries = [(c.kernel.value, c.cin_p, c.cout_p) for c in convs]
        if self._conv_batch is None or not self._conv_batch.matches(entries, self.dtype):
            self._conv_batch = ops.WeightPairBatch(entries, self.dtype)
        for c, (w_t, w_d) in zip(convs, self._conv_batch.run()):
            c.w_t, c.w_d = w_t, w_d
            c._version = self.weights_version
            c._dtype = self.dtype

    def wgrad_stream(self):
        """Side stream for the weight-gradient kernels (SIMCLR_WGRAD_STREAM=1), else None."""
        if self._wgrad_stream is None:
            import os
            on = os.environ.get('SIMCLR_WGRAD_STREAM', '0') not in ('', '0') and torch.cuda.is_available()
            self._wgrad_stream = torch.cuda.Stream() if on else False
        return self._wgrad_stream or None

    def unique(self, base):
        i = self.counters.get(base, 0)
        self.counters[base] = i + 1
        return base if i == 0 else '%s_%d' % (base, i)

    def path(self, *leaves):
        return '/'.join(self.scope + list(leaves))

    @property
    def dtype(self):
        return torch.bfloat16 if FLAGS.compute_dtype == 'bf16' else torch.float32


RT = _Runtime()


class scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        RT.scope.append(self.name)

    def __exit__(self, *a):
        RT.scope.pop()


def pad64(c):
    """Channel counts that are not a multiple of the 64-element k-tile are zero-padded internally."""
    return (c + 63) // 64 * 64


class Act:
    """An activation tensor plus (optionally) the fused per-channel statistics partials.  `c` is the
    logical channel count when the tensor carries zero-padded channels (t.shape[-1] = pad64(c))."""
    __slots__ = ('t', 'stats', 'c', 'shape', 'sums')

    def __init__(self, t, stats=None, c=None, shape=None, sums=None):
        # t is None for a statistics-only convolution pass (the tensor is never materialised): shape tells its geometry;
        # sums: this replica's [2, C] fp64 (sum, sum of squares) when the statistics did not come as partial slots
        self.t = t
        self.stats = stats
        self.sums = sums
        self.shape = tuple(t.shape) if t is not None else tuple(shape)
        self.c = self.shape[-1] if c is None else c


class PackedInput:
    """The k views of the input batch, packed by simclr_pack_views for the stem conv."""

    def __init__(self, images, num_views, kernel_size, strides, dtype, presplit_for_cout=0):
        b, H, W, _ = images.shape
        self.geo = ops.stem_geometry(H, W, kernel_size, kernel_size, strides)
        self.H, self.W = H, W
        self.xq = None
        if presplit_for_cout and dtype == torch.float32 and ops.stem_wgrad_ps_supported(self.geo, kernel_size, strides, presplit_for_cout):
            # the stem's weight gradient reads the image as bf16 pieces (simclr_stem_wgrad_ps): written in the packing pass
            self.xp, self.xq = ops.pack_views(images, num_views, self.geo, dtype, with_presplit=True)
        else:
            self.xp = ops.pack_views(images, num_views, self.geo, dtype)
        self.V = num_views * b


class Layer:
    trainable = True

    def sublayers(self):
        out = []
        for v in self.__dict__.values():
            if isinstance(v, Layer):
                out.append(v)
            elif isinstance(v, (list, tuple)):
                out.extend(x for x in v if isinstance(x, Layer))
        return out

    @property
    def variables(self):
        vs = [v for v in self.__dict__.values() if isinstance(v, Variable)]
        for l in self.sublayers():
            vs.extend(l.variables)
        return vs

    @property
    def trainable_variables(self):
        return [v for v in self.variables if v.trainable]


def _variance_scaling(shape, fan_in, gen):
    """tf.keras.initializers.VarianceScaling() defaults (tf2/resnet.py:201): truncated normal,
    stddev = sqrt(1/fan_in)/.87962566103423978, resampled outside 2 sigma."""
    std = math.sqrt(1.0 / fan_in) / .87962566103423978
    w = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return w * std


def _gen():
    RT.seed += 1
    return torch.Generator().manual_seed(RT.seed)


# --------------------------------------------------------------------------- BatchNormRelu
class BatchNormRelu(Layer):  # tf2/resnet.py:31-78
    def __init__(self, relu=True, init_zero=False, center=True, scale=True,
                 data_format='channels_last', **kwargs):
        if data_format != 'channels_last':
            raise ValueError('MI355X build supports channels_last only')
        self.relu = relu
        self.init_zero = init_zero
        self.center = center
        self.scale = scale
        self.trainable = kwargs.get('trainable', True)
        outer = RT.unique('batch_norm_relu')
        bn = RT.unique('sync_batch_normalization' if FLAGS.global_bn else 'batch_normalization')
        self._base = RT.path(outer, bn)
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None
        self.saved = None

    def build(self, C, Cp=None):
        """C logical channels; Cp >= C channels of the (zero-padded) tensor.  Storage is padded, the
        Variables are views of the first C entries; pad gamma/beta are 0 so pad channels stay 0."""
        dev = RT.device
        Cp = C if Cp is None else Cp
        self._pg = self._pb = None
        if self.scale:
            self._pg = torch.zeros(Cp, device=dev)
            if not self.init_zero:                                       # :42-45
                self._pg[:C] = 1.0
            self.gamma = Variable(self._base + '/gamma:0', self._pg[:C], self.trainable)
        if self.center:
            self._pb = torch.zeros(Cp, device=dev)
            self.beta = Variable(self._base + '/beta:0', self._pb[:C], self.trainable)
        self._pmm = torch.zeros(Cp, device=dev)
        self._pmv = torch.ones(Cp, device=dev)
        self.moving_mean = Variable(self._base + '/moving_mean:0', self._pmm[:C], False)
        self.moving_variance = Variable(self._base + '/moving_variance:0', self._pmv[:C], False)

    def prepare(self, inputs, training):
        """Statistics -> (mean, rstd, scale, shift); moving-average update when training.  If `prepare_many` already
        handled this input (statistics exchanged together with another layer's), its result is used."""
        p = getattr(self, '_prep', None)
        if p is not None and p[0] is inputs.t:
            self._prep = None
            return p[1], p[2]
        self._prep = None
        return prepare_many([(self, inputs)], training)[0]

    def _prepare_with(self, inputs, training, sums):
        """sums: [2,C] fp64 sums -- cross-replica ones under SyncBatchNormalization (:50-60, prepare_many all-reduced
        them), this replica's own otherwise (Gram-matrix statistics of a fused tail) -- or None (the slot reduction is fused
        into the finalize kernel).  The divisor follows the same rule: global rows only when the sums are global."""
        x = inputs.t
        C = inputs.shape[-1]
        if self.moving_mean is None:
            self.build(inputs.c, C)
        g, b = self._pg, self._pb
        rows = math.prod(inputs.shape[:-1])
        if training:
            if inputs.stats is None and sums is None:
                raise NotImplementedError('BatchNormRelu input must come from a conv/dense epilogue')
            if sums is not None:
                count = rows * (num_replicas(RT.strategy) if _sync_bn() else 1)
                mean, rstd, scale, shift = ops.bn_finalize(sums, count, g, b, self._pmm, self._pmv,
                                                           FLAGS.batch_norm_decay, BATCH_NORM_EPSILON)
            else:
                count = rows
                mean, rstd, scale, shift = ops.bn_finalize(None, count, g, b, self._pmm, self._pmv,
                                                           FLAGS.batch_norm_decay, BATCH_NORM_EPSILON,
                                                           partial=inputs.stats)
        else:
            mean = self._pmm
            rstd = torch.rsqrt(self._pmv + BATCH_NORM_EPSILON)
            scale = rstd if g is None else g * rstd
            shift = -mean * scale if b is None else b - mean * scale
            count = rows
        self.saved = dict(x=x, mean=mean, rstd=rstd, scale=scale, shift=shift, count=count, y=None, c=inputs.c)
        return scale, shift

    def __call__(self, inputs, training, relu=None, add=None, add_bn=None, want_bits=False):
        """BN (+ReLU).  `add` (a tensor) / `add_bn` ((scale, shift) of a projection shortcut) fuse
        the residual tail relu(bn(x)+shortcut) of resnet.py:382,487 into the same pass.  want_bits: also
        keep the ReLU mask as a bit tensor (self.relu_bits) for the fused backward of the tail."""
        relu = self.relu if relu is None else relu
        scale, shift = self.prepare(inputs, training)
        rs, rb = add_bn if add_bn is not None else (None, None)
        self.relu_bits = None
        if want_bits and relu:
            y, self.relu_bits = ops.bn_apply(inputs.t, scale, shift, relu, res=add, rscale=rs, rshift=rb, want_bits=True)
        else:
            y = ops.bn_apply(inputs.t, scale, shift, relu, res=add, rscale=rs, rshift=rb)
        self.saved['y'] = y
        self.saved['masked'] = bool(relu)
        return Act(y, c=inputs.c)

    def fusion_info(self, mask_src=None, sums_only=False):
        """What a consumer conv's dgrad epilogue needs to fuse this layer's backward reduce
        (simclr_conv2d_dgrad_bn): plain BN+ReLU -> mask recomputed from x (mode 2); residual tail
        -> mask from the block output (mode 1)."""
        s = self.saved
        if mask_src is not None:
            if getattr(self, 'relu_bits', None) is not None:    # 1 bit per element instead of the whole tensor
                if sums_only:
                    # the consumer produces sum(dm) only; sum(dm*x^) follows from the weight-gradient GEMM of the conv that
                    # feeds this BN (Conv2dFixedPadding.backward_folded) -- this BN's input is not read at all
                    return dict(mask=self.relu_bits, mode=4)
                return dict(x=s['x'], mask=self.relu_bits, mean=s['mean'], rstd=s['rstd'], mode=3)
            return dict(x=s['x'], mask=mask_src, mean=s['mean'], rstd=s['rstd'], mode=1)
        assert s.get('masked'), 'fusion_info without mask_src needs a BN+ReLU layer'
        return dict(x=s['x'], scale=s['scale'], shift=s['shift'], mean=s['mean'], rstd=s['rstd'], mode=2)

    def backward_fused(self, dm, partial, coeffs=None, ps_for=None):
        """Second half of the backward when the reduce was fused into the producing dgrad: dm is the
        already-masked gradient, partial the per-channel (sum dm, sum dm*x^) slots.  coeffs: (c1, c2) if
        `bwd_finalize_many` already exchanged / finalised the sums together with another layer's.
        ps_for: the convolution whose backward is the ONLY consumer of the result -- if its GEMMs take it, dx is written in the
        pre-split block format (fp32 parity mode: no operand splitting in their k-loops)."""
        s = self.saved
        c1, c2 = coeffs if coeffs is not None else self._bwd_finalize(partial, s['count'])
        dx, _ = ops.bn_bwd_apply(dm, s['x'], None, s['scale'], s['shift'], s['mean'], s['rstd'], c1, c2, 0,
                                 ps_out=_ps_grad_ok(ps_for))
        self.saved = None
        return dx

    def _bwd_finalize(self, partial, count):
        """partial slots -> (dgamma, dbeta written) and the coefficients c1 = mean(dy), c2 = mean(dy*x^).
        dgamma/dbeta take the LOCAL sums (the gradient all-reduce sums them), c1/c2 the GLOBAL ones."""
        return bwd_finalize_many([(self, partial)])[0]

    def _bwd_finalize_with(self, partial, local, glob):
        dgamma = self.gamma.ensure_grad() if self.gamma is not None and self.gamma.trainable else None
        dbeta = self.beta.ensure_grad() if self.beta is not None and self.beta.trainable else None
        count = self.saved['count']
        if local is not None:
            return ops.bn_bwd_finalize(local, glob, count, dgamma, dbeta)
        return ops.bn_bwd_finalize(None, None, count, dgamma, dbeta, partial=partial)

    def finalize_from_sums(self, local):
        """local: [2, C] fp64 (sum dm, sum dm*x^) of THIS replica -> (c1, c2); dgamma / dbeta from the local sums, the
        coefficients from the cross-replica sums (SyncBN collective C)."""
        dgamma = self.gamma.ensure_grad() if self.gamma is not None and self.gamma.trainable else None
        dbeta = self.beta.ensure_grad() if self.beta is not None and self.beta.trainable else None
        glob = RT.strategy.all_reduce_sum(local.clone()) if _sync_bn() else local
        return ops.bn_bwd_finalize(local, glob, self.saved['count'], dgamma, dbeta)

    def bwd_reduce(self, dy, mask_src=None, mask_mode=0):
        """First half of the un-fused backward: per-channel (sum dy_m, sum dy_m * x^) partial slots."""
        s = self.saved
        return ops.bn_bwd_reduce(dy, s['x'], mask_src, s['scale'], s['shift'], s['mean'], s['rstd'], mask_mode)

    def backward(self, dy, mask_src=None, mask_mode=None, want_masked=False, coeffs=None, ps_for=None):
        """dy: gradient wrt this layer's (activated) output.  Returns (dx, dy_masked).
        coeffs: (c1, c2) when the reduce + statistic exchange was already done (bwd_reduce + bwd_finalize_many).
        ps_for: see backward_fused."""
        s = self.saved
        if mask_mode is None:
            # plain BN+ReLU: the ReLU mask is recomputed from x*scale+shift (exactly (y > 0)),
            # which saves re-reading the activated output in both backward passes
            mask_mode = 2 if s.get('masked') else 0
            mask_src = None
        x = s['x']
        if coeffs is None:
            part = ops.bn_bwd_reduce(dy, x, mask_src, s['scale'], s['shift'], s['mean'], s['rstd'], mask_mode)
            c1, c2 = self._bwd_finalize(part, s['count'])
        else:
            c1, c2 = coeffs
        dx, dmasked = ops.bn_bwd_apply(dy, x, mask_src, s['scale'], s['shift'], s['mean'], s['rstd'], c1, c2,
                                       mask_mode, want_masked=want_masked, ps_out=_ps_grad_ok(ps_for))
        self.saved = None
        return dx, dmasked


def _ps_grad_ok(conv):
    """May the gradient handed to `conv.backward` be in the pre-split block format (csrc/common.h)?  fp32 storage with three bf16
    backward terms, a plain (not stem, not channel-padded) convolution on the LDS-DMA weight-gradient path."""
    if conv is None or RT.dtype != torch.float32 or not isinstance(conv, Conv2dFixedPadding):
        return False
    sv = conv.saved
    if sv is not None and 'packed' in sv:
        # the stem: its weight gradient is the only consumer (no data gradient); simclr_stem_wgrad_ps covers the 7x7 / stride-2 stem
        pk = sv['packed']
        return (conv.kernel is not None and conv.kernel.trainable and conv.cout_p == conv.filters
                and ops.stem_wgrad_ps_supported(pk.geo, conv.kernel_size, conv.strides, conv.cout_p))
    if sv is None or conv.kernel is None or conv.padded or not conv.kernel.trainable:
        return False
    return conv.cin_p % 64 == 0 and conv.cout_p % 32 == 0 and ops.ps_backward_enabled()


def _conv3_fused_level():
    """SIMCLR_CONV3_FUSED: 0 = every bottleneck tail runs as conv3 -> HBM -> bn_apply; 1 = identity blocks use the fused
    forward; 2 (default) = projection blocks too (the shortcut's own BatchNorm is applied in the same epilogue): with the
    Gram-matrix statistics 0.5-0.7 ms/step faster than 1 in three interleaved pairs (profiles/r02_notes.md)."""
    import os
    v = os.environ.get('SIMCLR_CONV3_FUSED', '2')
    return int(v) if v.isdigit() else 2


def _conv3_stats_from_gram():
    """SIMCLR_CONV3_STATS=conv: the statistics pass of the fused tail is a store-free run of the convolution (bitwise the
    statistics of the unfused path); default: from the Gram matrix of conv3's input (shared with the backward)."""
    import os
    return os.environ.get('SIMCLR_CONV3_STATS', 'gram') != 'conv'


def _sync_bn():
    return FLAGS.global_bn and collectives_on(RT.strategy)


def prepare_many(items, training):
    """items: [(BatchNormRelu, Act)] whose inputs do not depend on each other (a projection shortcut's BN and bn1 of
    the same block).  Cross-replica statistics (tf2/resnet.py:50-60) of all of them travel in ONE all-reduce.
    Returns [(scale, shift)] and leaves each result cached on its layer for the following __call__ / prepare."""
    sums = [a.sums if training else None for _, a in items]
    if training and _sync_bn():
        sums = RT.strategy.all_reduce_sum_many([a.sums.clone() if a.sums is not None else ops.bn_reduce_slots(a.stats)
                                                for _, a in items])
    out = []
    for (bn, a), sm in zip(items, sums):
        scale, shift = bn._prepare_with(a, training, sm)
        bn._prep = (a.t, scale, shift)
        out.append((scale, shift))
    if len(items) == 1:
        items[0][0]._prep = None
    return out


def bwd_finalize_many(items):
    """items: [(BatchNormRelu, partial slots)] of independent backward reductions (a residual tail's BN and the
    projection shortcut's BN share the same upstream gradient).  One all-reduce for all of them.  Returns [(c1, c2)]."""
    if _sync_bn():
        local = [ops.bn_reduce_slots(p) for _, p in items]
        glob = RT.strategy.all_reduce_sum_many([l.clone() for l in local])
        return [bn._bwd_finalize_with(p, l, g) for (bn, p), l, g in zip(items, local, glob)]
    return [bn._bwd_finalize_with(p, None, None) for bn, p in items]


# --------------------------------------------------------------------------- convolution
class FixedPadding(Layer):  # tf2/resnet.py:160-180 -- folded into the conv kernels' gather
    def __init__(self, kernel_size, data_format='channels_last', **kwargs):
        self.kernel_size = kernel_size


class Conv2dFixedPadding(Layer):  # tf2/resnet.py:183-208
    def __init__(self, filters, kernel_size, strides, data_format='channels_last', **kwargs):
        if data_format != 'channels_last':
            raise ValueError('MI355X build supports channels_last only')
        self.filters = filters
        self.kernel_size = kernel_size
        self.strides = strides
        self.trainable = kwargs.get('trainable', True)
        self._name = self._make_name()
        self.kernel = None
        self._version = -1
        self.saved = None

    def _make_name(self):
        outer = RT.unique('conv2d_fixed_padding')
        inner = RT.unique('conv2d')
        return RT.path(outer, inner, 'kernel:0')

    def build(self, cin, cin_p=None):
        """cin logical input channels; cin_p channels of the (zero-padded) input tensor."""
        k = self.kernel_size
        w = _variance_scaling((k, k, cin, self.filters), k * k * cin, _gen())
        self.kernel = Variable(self._name, w.to(RT.device), self.trainable)
        self.cin = cin
        self.cin_p = cin if cin_p is None else cin_p
        self.cout_p = pad64(self.filters) if self.filters % 64 else self.filters
        self.padded = self.cin_p != cin or self.cout_p != self.filters
        if not getattr(self, 'is_stem', False) and cin > 4:
            RT.convs.append(weakref.ref(self))

    def _refresh(self, stem_geo=None):
        if self._version == RT.weights_version and getattr(self, '_dtype', None) == RT.dtype:
            return
        w = self.kernel.value
        if stem_geo is not None:
            self.w_s = ops.prep_weights(w, 2, RT.dtype, stem_geo['KHP'], stem_geo['KWP'], cout_p=self.cout_p)
        else:
            # first refresh of a layer (lazy build during the first forward) or a layer outside the registry (built under
            # an earlier RT.reset()): its own launch; afterwards every version bump refreshes ALL layers with one launch
            if self._version < 0 or not any(r() is self for r in RT.convs):
                self.w_t, self.w_d = ops.prep_weights_pair(w, RT.dtype, cin_p=self.cin_p, cout_p=self.cout_p)
            else:
                RT.refresh_conv_weights()
                return
        self._version = RT.weights_version
        self._dtype = RT.dtype

    def __call__(self, inputs, training, want_stats=True):
        k, s = self.kernel_size, self.strides
        if isinstance(inputs, PackedInput):
            if self.kernel is None:
                self.build(3)
            self._refresh(inputs.geo)
            stats = (ops.stem_stats(inputs.V * inputs.geo['OH'] * inputs.geo['OW'], self.cout_p, RT.device)
                     if (want_stats and training) else None)
            y = ops.stem_conv_fwd(inputs.xp, self.w_s, inputs.geo, s, stats=stats)
            self.saved = dict(packed=inputs)
            return Act(y, stats, c=self.filters)
        x = inputs.t
        V, H, W, cin_p = x.shape
        if self.kernel is None:
            self.build(inputs.c, cin_p)
        self._refresh()
        pad = (k - 1) // 2                                  # FixedPadding / SAME at stride 1
        OH = (H + (k - 1) - k) // s + 1
        OW = (W + (k - 1) - k) // s + 1
        stats = ops.conv_stats(V * OH * OW, self.cout_p, RT.device) if (want_stats and training) else None
        # fp32: statistics about a per-channel pivot, handed on as this replica's fp64 moments (Act.sums)
        y, stats, sums = ops.conv2d_fwd_with_stats(x, self.w_t, k, k, s, pad, OH, OW, stats)
        self.saved = dict(x=x, H=H, W=W, pad=pad)
        return Act(y, stats, c=self.filters, sums=sums)

    def _store_wgrad(self, tmp4):
        """tmp4: [k, k, cin_p, cout_p] fp32 -> logical slice into the gradient buffer."""
        g = self.kernel.ensure_grad()
        g.copy_(tmp4[:, :, :g.shape[2], :self.filters])

    def _f32_copies(self):
        """fp32 views of the compute copies (the values the forward multiplied with), cached per weight version."""
        if getattr(self, '_v32', -1) != self._version:
            self.w_d32 = self.w_d if self.w_d.dtype == torch.float32 else ops.cast(self.w_d, torch.float32)
            self.w_t32 = self.w_t if self.w_t.dtype == torch.float32 else ops.cast(self.w_t, torch.float32)
            self._v32 = self._version
        return self.w_d32, self.w_t32

    def forward_stats_only(self, inputs):
        """First half of the fused conv + BatchNorm-apply forward: the statistics of this convolution's output, which is
        not stored.  Saves the input for the backward like __call__."""
        k, s = self.kernel_size, self.strides
        x = inputs.t
        V, H, W, cin_p = x.shape
        if self.kernel is None:
            self.build(inputs.c, cin_p)
        self._refresh()
        pad = (k - 1) // 2
        OH = (H + (k - 1) - k) // s + 1
        OW = (W + (k - 1) - k) // s + 1
        stats = ops.conv_stats(V * OH * OW, self.cout_p, RT.device)
        ops.conv2d_fwd(x, self.w_t, k, k, s, pad, OH, OW, stats=stats, store=False)
        self.saved = dict(x=x, H=H, W=W, pad=pad)
        return Act(None, stats, c=self.filters, shape=(V, OH, OW, self.cout_p))

    def forward_gram_stats(self, inputs):
        """Statistics of this 1x1 convolution's output from the Gram matrix of its INPUT (csrc/bn.hip bn_sums_from_gram):
        one pass over the input, 4x narrower than the output, and h^T h, colsum(h), (h^T h) W are exactly what the folded
        BatchNorm backward of this layer needs -- kept in self.saved['gram'] so that the backward does not stream the
        input again."""
        assert self.kernel_size == 1 and self.strides == 1
        x = inputs.t
        V, H, W, cin_p = x.shape
        if self.kernel is None:
            self.build(inputs.c, cin_p)
        self._refresh()
        K = cin_p
        if ops.gram_supported(K, x.dtype):
            g, cs = ops.conv2d_gram(x)
        else:
            g = ops.conv2d_wgrad(x, x, 1, 1, 1, 0)
            cs = ops.bn_reduce_slots(ops.bn_bwd_reduce(x, x, None, None, None, RT.const(K, 0.0), RT.const(K, 1.0), 0))
        w_d32, w_t32 = self._f32_copies()
        gw = ops.small_gemm_nt(g, w_t32)                       # (h^T h) W   [K, N]
        sums = ops.bn_sums_from_gram(gw, w_d32, cs)
        self.saved = dict(x=x, H=H, W=W, pad=0, gram=(g, cs, gw), gram_version=self._version)
        return Act(None, None, c=self.filters, shape=(V, H, W, self.cout_p), sums=sums)

    def forward_bn_apply(self, inputs, scale, shift, res=None, relu=True, want_bits=False, res_bn=None):
        """Second half: the convolution again, with y = act(bn(conv) + res) applied in its epilogue."""
        k, s = self.kernel_size, self.strides
        x = inputs.t
        V, H, W, _ = x.shape
        pad = (k - 1) // 2
        OH = (H + (k - 1) - k) // s + 1
        OW = (W + (k - 1) - k) // s + 1
        rs, rb = res_bn if res_bn is not None else (None, None)
        return ops.conv2d_fwd_bn_apply(x, self.w_t, k, k, s, pad, OH, OW, scale, shift, res=res, relu=relu,
                                       want_bits=want_bits, rscale=rs, rshift=rb)

    def backward_folded(self, dm, bn_out, partial, fuse_bn, s2_from_gemm=False, dx_out=None, accumulate=False):
        """1x1 stride-1 conv whose output c = h W goes through `bn_out` (BatchNorm, no ReLU before the add): the BN backward
        dh = a*dm + b*c + d is folded into this layer's gradients by linearity (csrc/bn.hip bn_fold_*), so neither the
        streaming BN-backward pass nor dh exists.  dm: masked gradient wrt bn_out's output; coeffs = (c1, c2) of bn_out.
        Returns (dm_in, partial) like backward(..., fuse_bn=...)."""
        assert self.kernel_size == 1 and self.strides == 1 and not self.padded
        sv = self.saved
        self.saved = None
        h = sv['x']
        V, H, W, K = h.shape
        N = self.cout_p
        st = bn_out.saved
        t1 = ops.conv2d_wgrad(h, dm, 1, 1, 1, 0)                                                # h^T dm      [K, N]
        if s2_from_gemm:
            # `partial` carries sum(dm) only (dgrad epilogue mode 4): sum(dm*x^) = rstd * (<W, T1>_k - mean * sum(dm))
            local = ops.bn_reduce_slots(partial)
            ops.bn_fold_s2(t1, self.w_d, st['mean'], st['rstd'], local)
            coeffs = bn_out.finalize_from_sums(local)
        else:
            coeffs = bn_out._bwd_finalize(partial, st['count'])
        a, b, d, wb, wext, e = ops.bn_fold_pre(self.w_d, st['scale'], st['mean'], st['rstd'], coeffs[0], coeffs[1])
        w_d32, w_t32 = self._f32_copies()
        q = ops.small_gemm_nt(wb, w_d32)                                                        # (W*b) W^T   [K, K]
        if self.kernel.trainable:
            with _wgrad_side_stream(h, dm):
                if sv.get('gram') is not None and sv.get('gram_version') == self._version:
                    g, cs, gw = sv['gram']                  # the forward's statistics pass already produced all three
                else:
                    if ops.gram_supported(K, h.dtype):
                        g, cs = ops.conv2d_gram(h)                                              # h^T h, colsum(h): ONE pass over h
                    else:
                        g = ops.conv2d_wgrad(h, h, 1, 1, 1, 0)
                        cs = ops.bn_reduce_slots(ops.bn_bwd_reduce(h, h, None, None, None, RT.const(K, 0.0), RT.const(K, 1.0), 0))
                    gw = ops.small_gemm_nt(g, w_t32)                                            # (h^T h) W   [K, N]
                ops.bn_fold_post(t1, gw, cs, a, b, d, q, self.kernel.ensure_grad().view(K, N), wext)
        else:
            z = torch.zeros(K, N, device=h.device)
            ops.bn_fold_post(z, z, torch.zeros(K, device=h.device), a, b, d, q, torch.empty(K, N, device=h.device), wext)
        join_wgrad_stream()                       # wext's last K columns come from bn_fold_post
        if fuse_bn is None:                       # conv input is not a BatchNorm output (projection shortcut at a block entry)
            return ops.conv2d_dgrad_ext(dm, h, wext, e, out=dx_out, accumulate=accumulate), None
        return ops.conv2d_dgrad_bn_ext(dm, h, wext, e, fuse_bn)

    def backward(self, dy, need_dx=True, dx_out=None, accumulate=False, fuse_bn=None, sparse=False):
        """Returns dx, or (dm, partial) when `fuse_bn` (BatchNormRelu.fusion_info of the layer that
        produced this conv's input) asks for the fused BN-backward reduce (stride-1 convs only).
        sparse: see ops.conv2d_dgrad (a stride-2 1x1 convolution whose dx an accumulating data gradient completes)."""
        k, s = self.kernel_size, self.strides
        sv = self.saved
        self.saved = None
        train_w = self.kernel.trainable
        if 'packed' in sv:
            if train_w:
                pk = sv['packed']
                with _wgrad_side_stream(pk.xp, dy):      # same stream as every other wgrad: they share one workspace
                    if self.cout_p == self.filters:
                        ops.stem_conv_wgrad(pk.xp, dy, pk.geo, k, k, s, out=self.kernel.ensure_grad(), xq=pk.xq)
                    else:
                        self._store_wgrad(ops.stem_conv_wgrad(pk.xp, dy, pk.geo, k, k, s, xq=pk.xq))
            return None
        if train_w:
            with _wgrad_side_stream(sv['x'], dy):
                if not self.padded:
                    ops.conv2d_wgrad(sv['x'], dy, k, k, s, sv['pad'], out=self.kernel.ensure_grad().view(-1, self.filters))
                else:
                    tmp = ops.conv2d_wgrad(sv['x'], dy, k, k, s, sv['pad'])
                    self._store_wgrad(tmp.view(k, k, self.cin_p, self.cout_p))
        if not need_dx:
            return None
        if fuse_bn is not None:
            assert s == 1
            return ops.conv2d_dgrad_bn(dy, self.w_d, k, k, sv['pad'], sv['H'], sv['W'], fuse_bn, out=dx_out,
                                       accumulate=accumulate)
        return ops.conv2d_dgrad(dy, self.w_d, k, k, s, sv['pad'], sv['H'], sv['W'], out=dx_out,
                                accumulate=accumulate, sparse=sparse)


def _pool_fusion_enabled():
    """Max-pool backward fused into the stem BatchNorm's backward (the un-pooled gradient is never written or re-read).  Default: on in
    fp32 storage (round 6, with the four windows of a pixel requested up front: 3.28 against 3.94 ms for the three unfused kernels,
    -0.8 ms per step in three interleaved pairs), off in bf16 (round 2: 0.4 ms per step slower, profiles/r02_notes.md);
    SIMCLR_POOL_FUSION=0 | 1 overrides."""
    import os
    e = os.environ.get('SIMCLR_POOL_FUSION')
    if e is not None and e != '':
        return e != '0'
    return RT.dtype == torch.float32


def _bn_s2_enabled():
    import os
    return os.environ.get('SIMCLR_BN_S2_GEMM', '1') not in ('', '0')


def _bn_fold_enabled():
    import os
    return os.environ.get('SIMCLR_BN_FOLD', '1') not in ('', '0')


class _wgrad_side_stream:
    """Weight gradients have no consumer until the optimizer step, while the data gradient of the same layer is on the
    critical path of the backward pass.  With SIMCLR_WGRAD_STREAM=1 every wgrad launch goes to a second HIP stream
    (after an event that marks its operands ready), so it runs concurrently with the dgrad / BatchNorm-backward chain;
    `join_wgrad_stream()` makes the launch stream wait for it before gradients are reduced / applied."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        ws = RT.wgrad_stream()
        if ws is None:
            return self
        ev = torch.cuda.Event()
        ev.record()
        ws.wait_event(ev)
        for t in self.tensors:
            if t is not None:
                t.record_stream(ws)          # the caching allocator must not recycle them before the side stream is done
        self.ctx = torch.cuda.stream(ws)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


def join_wgrad_stream():
    ws = RT.wgrad_stream()
    if ws is not None:
        torch.cuda.current_stream().wait_stream(ws)


class _PlainConv1x1(Conv2dFixedPadding):
    """The bare tf.keras.layers.Conv2D(k=1) layers inside SK_Conv2D (tf2/resnet.py:243-256)."""

    def __init__(self, filters):
        super().__init__(filters, 1, 1)

    def _make_name(self):
        return RT.path(RT.unique('conv2d'), 'kernel:0')


class IdentityLayer(Layer):  # tf2/resnet.py:211-214
    def __init__(self, name=None, **kwargs):
        self.name = name

    def __call__(self, inputs, training):
        return inputs


def _no_dropblock(keep_prob, size):
    if keep_prob is not None and keep_prob != 1.0:
        raise NotImplementedError('DropBlock (tf2/resnet.py:81-157) is not part of the hot path')


# --------------------------------------------------------------------------- blocks
class SK_Conv2D(Layer):  # pylint: disable=invalid-name
    """Selective kernel convolutional layer (tf2/resnet.py:217-277): a 3x3 conv producing two streams
    (2f channels), BN+ReLU, a squeeze path (global mean of the stream sum -> 1x1 -> BN+ReLU -> 1x1)
    whose 2-way softmax mixes the streams."""

    def __init__(self, filters, strides, sk_ratio, min_dim=32, data_format='channels_last', **kwargs):
        self.filters = filters
        self.sk_ratio = sk_ratio
        self.min_dim = min_dim
        with scope(RT.unique('sk__conv2d')):
            self.conv2d_fixed_padding = Conv2dFixedPadding(filters=2 * filters, kernel_size=3, strides=strides,
                                                           data_format=data_format)          # :231-236
            self.batch_norm_relu = BatchNormRelu(data_format=data_format)                     # :237
            mid_dim = max(int(filters * sk_ratio), min_dim)                                    # :242
            self.conv2d_0 = _PlainConv1x1(mid_dim)                                             # :243-249
            self.batch_norm_relu_1 = BatchNormRelu(data_format=data_format)                   # :250
            self.conv2d_1 = _PlainConv1x1(2 * filters)                                         # :251-257
        self.saved = None

    @property
    def strides(self):
        return self.conv2d_fixed_padding.strides

    def __call__(self, inputs, training):
        f = self.filters
        a = self.batch_norm_relu(self.conv2d_fixed_padding(inputs, training), training)       # :264-265
        V = a.t.shape[0]
        g = ops.sk_pool_fwd(a.t, f, pad64(f))                                                 # :266-270
        h = self.conv2d_0(Act(g.view(V, 1, 1, -1), c=f), training)                            # :271
        h = self.batch_norm_relu_1(h, training)                                               # :272
        l = self.conv2d_1(h, training, want_stats=False)                                      # :273
        lt = l.t.view(V, -1)
        out = ops.sk_mix_fwd(a.t, lt, f)                                                      # :274-277
        self.saved = dict(a=a.t, l=lt)
        return Act(out, c=f)

    def backward(self, dout, fuse_bn=None):
        """dout: gradient wrt the mixed output [V,H,W,f].  Returns what the 3x3 conv's backward returns."""
        f = self.filters
        sv = self.saved
        self.saved = None
        a, l = sv['a'], sv['l']
        V = a.shape[0]
        dl = ops.sk_mix_bwd_logits(a, l, dout, f)
        dh = self.conv2d_1.backward(dl.view(V, 1, 1, -1))
        dh, _ = self.batch_norm_relu_1.backward(dh)
        dg = self.conv2d_0.backward(dh).view(V, -1)
        da = ops.sk_mix_bwd_streams(l, dout, dg, f)
        dconv, _ = self.batch_norm_relu.backward(da)
        return self.conv2d_fixed_padding.backward(dconv, fuse_bn=fuse_bn)


class _Shortcut(Layer):
    """Projection shortcut (tf2/resnet.py:328-349, 398-423): Conv2dFixedPadding 1x1 (stride s) +
    BatchNormRelu(relu=False); with sk_ratio>0 the ResNet-D form: [FixedPadding(2)] ->
    AveragePooling2D(2, s) -> 1x1 conv stride 1.  The BN *apply* is deferred into the block's fused tail."""

    def __init__(self, filters_out, strides, data_format):
        self.resnet_d = FLAGS.sk_ratio > 0
        self.strides = strides
        self.conv = Conv2dFixedPadding(filters=filters_out, kernel_size=1, strides=1 if self.resnet_d else strides,
                                       data_format=data_format)
        self.bn = BatchNormRelu(relu=False, data_format=data_format)

    def conv_part(self, inputs, training):
        """(avg-pool +) 1x1 conv; the BN statistics are exchanged by the caller together with bn1's."""
        if self.resnet_d:
            self._hw = inputs.t.shape[1:3]
            inputs = Act(ops.avgpool2_fwd(inputs.t, self.strides), c=inputs.c)
        return self.conv(inputs, training)

    def __call__(self, inputs, training):
        raw = self.conv_part(inputs, training)
        scale, shift = self.bn.prepare(raw, training)
        return raw.t, (scale, shift)

    def foldable(self):
        """Stride-1 1x1 projection (no avg-pool) with a supported channel count: its BN backward folds into the conv."""
        import os
        c = self.conv
        return (os.environ.get('SIMCLR_SC_FOLD', '1') not in ('', '0') and
                not self.resnet_d and c.strides == 1 and c.kernel is not None and not c.padded and
                ops.gram_supported(c.cin_p, RT.dtype) and _bn_fold_enabled() and _bn_s2_enabled())

    def backward_folded(self, d_sum, partial):
        """partial: slots whose sum(dm) part is valid for d_sum (the tail BN's, same upstream gradient)."""
        d, _ = self.conv.backward_folded(d_sum, self.bn, partial, None, s2_from_gemm=True)
        self.bn.saved = None
        return d

    def backward(self, d_sum, coeffs=None, sparse=False):
        """sparse: the caller completes the result with an accumulating data gradient (the block's conv1) -- a strided 1x1 projection
        then stores only the quarter of dx that receives a tap (ops.conv2d_dgrad sparse: no zero fill, nothing read back from it)."""
        d_raw, _ = self.bn.backward(d_sum, mask_mode=0, coeffs=coeffs, ps_for=self.conv)
        sp = (sparse and not self.resnet_d and self.conv.strides == 2 and not self.conv.padded and ops.sparse_dgrad_enabled(RT.dtype))
        d = self.conv.backward(d_raw, sparse=sp)
        if self.resnet_d:
            d = ops.avgpool2_bwd(d, self._hw[0], self._hw[1], self.strides)
        return d


def _block_entry(block, inputs, training):
    """Shortcut conv and conv1 read the same block input: run both, then ONE statistics exchange for the two
    BatchNorms (SyncBN collective C), then bn1's apply.  Returns (shortcut tensor, shortcut (scale, shift) | None, h1)."""
    if block.shortcut is None:
        return inputs.t, None, block.bn1(block.conv1(inputs, training), training)
    raw_sc = block.shortcut.conv_part(inputs, training)
    raw1 = block.conv1(inputs, training)
    (sc_bn, _) = prepare_many([(block.shortcut.bn, raw_sc), (block.bn1, raw1)], training)
    block.shortcut.bn._prep = None
    return raw_sc.t, sc_bn, block.bn1(raw1, training)


def _block_tail_backward(block, bn_tail, dout, dout_partial, conv_tail=None, sparse_shortcut=False):
    """Backward of relu(bn_tail(h) + shortcut): returns (dh, dx_shortcut_path).  When the tail's reduce arrived fused
    (dout_partial) and the block has a projection shortcut, the two BatchNorm backward reductions -- same upstream
    gradient -- share one statistics exchange."""
    if dout_partial is not None:
        dsum = dout
        if block.shortcut is not None:
            sc_part = block.shortcut.bn.bwd_reduce(dsum, mask_mode=0)
            co_t, co_s = bwd_finalize_many([(bn_tail, dout_partial), (block.shortcut.bn, sc_part)])
            dh = bn_tail.backward_fused(dout, dout_partial, coeffs=co_t, ps_for=conv_tail)
            return dh, block.shortcut.backward(dsum, coeffs=co_s, sparse=sparse_shortcut)
        dh = bn_tail.backward_fused(dout, dout_partial, ps_for=conv_tail)
    else:
        dh, dsum = bn_tail.backward(dout, mask_src=block.out, mask_mode=1, want_masked=True, ps_for=conv_tail)
    dx = block.shortcut.backward(dsum, sparse=sparse_shortcut) if block.shortcut is not None else dsum
    return dh, dx


class ResidualBlock(Layer):  # tf2/resnet.py:314-382
    def __init__(self, filters, strides, use_projection=False, data_format='channels_last',
                 dropblock_keep_prob=None, dropblock_size=None, **kwargs):
        del dropblock_keep_prob, dropblock_size          # :323-324
        if FLAGS.se_ratio > 0:
            raise NotImplementedError('SE_Layer (tf2/resnet.py:280-311) not built')
        with scope(RT.unique('residual_block')):
            self.shortcut = _Shortcut(filters, strides, data_format) if use_projection else None
            self.conv1 = Conv2dFixedPadding(filters=filters, kernel_size=3, strides=strides, data_format=data_format)
            self.bn1 = BatchNormRelu(data_format=data_format)
            self.conv2 = Conv2dFixedPadding(filters=filters, kernel_size=3, strides=1, data_format=data_format)
            self.bn2 = BatchNormRelu(relu=False, init_zero=True, data_format=data_format)

    def __call__(self, inputs, training):
        sc, sc_bn, h = _block_entry(self, inputs, training)
        h = self.conv2(h, training)
        out = self.bn2(h, training, relu=True, add=sc, add_bn=sc_bn, want_bits=training)     # relu(inputs + shortcut), :382
        self.out = out.t
        return out

    def tail_info(self):
        return self.bn2.fusion_info(mask_src=self.out)

    def backward(self, dout, dout_partial=None, prev_tail=None):
        """dout_partial given: dout is already ReLU-masked and the tail BN's reduce is done (fused into
        the next block's dgrad).  prev_tail: tail_info() of the block feeding this one -- its reduce is
        fused into this block's last dgrad.  Returns (dx, partial-or-None)."""
        dh, dx = _block_tail_backward(self, self.bn2, dout, dout_partial, self.conv2)
        self.out = None
        dm1, part1 = self.conv2.backward(dh, fuse_bn=self.bn1.fusion_info())
        dh1 = self.bn1.backward_fused(dm1, part1, ps_for=self.conv1)
        if prev_tail is not None and self.conv1.strides == 1:
            return self.conv1.backward(dh1, dx_out=dx, accumulate=True, fuse_bn=prev_tail)
        self.conv1.backward(dh1, dx_out=dx, accumulate=True)
        return dx, None


class BottleneckBlock(Layer):  # tf2/resnet.py:385-487
    def __init__(self, filters, strides, use_projection=False, data_format='channels_last',
                 dropblock_keep_prob=None, dropblock_size=None, **kwargs):
        _no_dropblock(dropblock_keep_prob, dropblock_size)
        if FLAGS.se_ratio > 0:
            raise NotImplementedError('SE_Layer (tf2/resnet.py:280-311) not built')
        with scope(RT.unique('bottleneck_block')):
            self.shortcut = _Shortcut(4 * filters, strides, data_format) if use_projection else None
            self.conv1 = Conv2dFixedPadding(filters=filters, kernel_size=1, strides=1, data_format=data_format)
            self.bn1 = BatchNormRelu(data_format=data_format)
            self.sk = None
            if FLAGS.sk_ratio > 0:                                                  # :442-444
                self.sk = SK_Conv2D(filters, strides, FLAGS.sk_ratio, data_format=data_format)
                self.conv2 = self.bn2 = None
            else:                                                                   # :446-453
                self.conv2 = Conv2dFixedPadding(filters=filters, kernel_size=3, strides=strides,
                                                data_format=data_format)
                self.bn2 = BatchNormRelu(data_format=data_format)
            self.conv3 = Conv2dFixedPadding(filters=4 * filters, kernel_size=1, strides=1, data_format=data_format)
            self.bn3 = BatchNormRelu(relu=False, init_zero=True, data_format=data_format)

    def __call__(self, inputs, training):
        sc, sc_bn, h = _block_entry(self, inputs, training)
        if self.sk is not None:
            h = self.sk(h, training)
        else:
            h = self.bn2(self.conv2(h, training), training)
        self.fused_tail = self._fused_tail(training, sc_bn, h)
        if self.fused_tail:
            # conv3's output is 4x wider than its input and only feeds bn3: it is never stored.  Its BatchNorm statistics
            # come from the Gram matrix of conv3's input (or a store-free run of the convolution), then the convolution runs
            # with relu(bn3(.) + shortcut) applied in its epilogue; the backward of this block never reads conv3's output
            # either (folded BatchNorm backward, sum(dm*x^) from the weight-gradient GEMM).
            st = self.conv3.forward_gram_stats(h) if _conv3_stats_from_gram() else self.conv3.forward_stats_only(h)
            scale, shift = self.bn3.prepare(st, training)
            y, bits = self.conv3.forward_bn_apply(h, scale, shift, res=sc, relu=True, want_bits=True, res_bn=sc_bn)
            self.bn3.relu_bits = bits
            self.bn3.saved['y'] = y
            self.bn3.saved['masked'] = True
            out = Act(y, c=st.c)
        else:
            h = self.conv3(h, training)
            out = self.bn3(h, training, relu=True, add=sc, add_bn=sc_bn, want_bits=training)     # relu(inputs + shortcut), :487
        self.out = out.t
        return out

    def _foldable(self):
        return self.sk is None and not self.conv3.padded and _bn_fold_enabled()

    def _fused_tail(self, training, sc_bn, h):
        # blocks whose tail BatchNorm backward will arrive folded (every block but the network's last one)
        if self.conv3.kernel is None:
            self.conv3.build(h.c, h.t.shape[-1])
        level = _conv3_fused_level()
        # fp32 storage (round 6): the same fusion where one Gram tile spans conv3's input channels (ops.gram_supported: K = 64 / 128,
        # i.e. the 56^2 and 28^2 blocks, which carry 3/4 of the tail bytes); SIMCLR_CONV3_FUSED_F32=0 keeps conv3 -> HBM -> bn_apply
        if RT.dtype != torch.bfloat16:
            import os
            if os.environ.get('SIMCLR_CONV3_FUSED_F32', '1') in ('', '0') or not _conv3_stats_from_gram():
                return False
            if not ops.gram_supported(self.conv3.cin_p, RT.dtype):
                return False
        return (training and not getattr(self, 'is_final', False)
                and self._foldable() and _bn_s2_enabled() and (level >= 2 or (level == 1 and sc_bn is None)))

    def tail_info(self):
        # foldable tail: the consumer's dgrad epilogue only masks and sums dm (no read of this block's conv3 output)
        return self.bn3.fusion_info(mask_src=self.out, sums_only=self._foldable() and _bn_s2_enabled())

    def backward(self, dout, dout_partial=None, prev_tail=None):
        """See ResidualBlock.backward.  Returns (dx, partial-or-None)."""
        fold = dout_partial is not None and self._foldable()
        if getattr(self, 'fused_tail', False) and not fold:
            raise RuntimeError('bottleneck block ran the fused conv3 + bn3 forward (conv3 output not stored) but its '
                               'backward did not receive the folded tail reduction; set SIMCLR_CONV3_FUSED=0')
        if fold:
            # tail BN3 backward folded into conv3's wgrad / dgrad (no bn_bwd_apply pass, no dh3 tensor); with
            # _bn_s2_enabled() the producer of `dout` did not even read conv3's output for the BN3 reduce
            if self.shortcut is None:
                dx = dout
            elif self.shortcut.foldable():
                dx = self.shortcut.backward_folded(dout, dout_partial)
            else:
                dx = self.shortcut.backward(dout, sparse=True)        # completed by conv1's accumulating data gradient below
            self.out = None
            dm2, part2 = self.conv3.backward_folded(dout, self.bn3, dout_partial, fuse_bn=self.bn2.fusion_info(),
                                                    s2_from_gemm=_bn_s2_enabled())
            self.bn3.saved = None
            dh2 = self.bn2.backward_fused(dm2, part2, ps_for=self.conv2)
            if self.conv2.strides == 1:
                dm1, part1 = self.conv2.backward(dh2, fuse_bn=self.bn1.fusion_info())
                dh1 = self.bn1.backward_fused(dm1, part1, ps_for=self.conv1)
            else:
                dh1, _ = self.bn1.backward(self.conv2.backward(dh2), ps_for=self.conv1)
            if prev_tail is not None:
                return self.conv1.backward(dh1, dx_out=dx, accumulate=True, fuse_bn=prev_tail)
            self.conv1.backward(dh1, dx_out=dx, accumulate=True)
            return dx, None
        dh3, dx = _block_tail_backward(self, self.bn3, dout, dout_partial, self.conv3, sparse_shortcut=True)
        self.out = None
        if self.sk is not None:
            dsk = self.conv3.backward(dh3)
            if self.sk.strides == 1:
                dm1, part1 = self.sk.backward(dsk, fuse_bn=self.bn1.fusion_info())
                dh1 = self.bn1.backward_fused(dm1, part1, ps_for=self.conv1)
            else:
                dh1, _ = self.bn1.backward(self.sk.backward(dsk), ps_for=self.conv1)
        else:
            dm2, part2 = self.conv3.backward(dh3, fuse_bn=self.bn2.fusion_info())
            dh2 = self.bn2.backward_fused(dm2, part2, ps_for=self.conv2)
            if self.conv2.strides == 1:
                dm1, part1 = self.conv2.backward(dh2, fuse_bn=self.bn1.fusion_info())
                dh1 = self.bn1.backward_fused(dm1, part1, ps_for=self.conv1)
            else:
                dh1, _ = self.bn1.backward(self.conv2.backward(dh2), ps_for=self.conv1)
        if prev_tail is not None:
            return self.conv1.backward(dh1, dx_out=dx, accumulate=True, fuse_bn=prev_tail)
        self.conv1.backward(dh1, dx_out=dx, accumulate=True)
        return dx, None


class BlockGroup(Layer):  # tf2/resnet.py:490-526
    def __init__(self, filters, block_fn, blocks, strides, data_format='channels_last',
                 dropblock_keep_prob=None, dropblock_size=None, **kwargs):
        self._name = kwargs.get('name')
        with scope(self._name):
            self.layers = [block_fn(filters, strides, use_projection=True, data_format=data_format,
                                    dropblock_keep_prob=dropblock_keep_prob, dropblock_size=dropblock_size)]
            for _ in range(1, blocks):
                self.layers.append(block_fn(filters, 1, data_format=data_format,
                                            dropblock_keep_prob=dropblock_keep_prob,
                                            dropblock_size=dropblock_size))

    def __call__(self, inputs, training):
        for layer in self.layers:
            inputs = layer(inputs, training)
        return inputs

    def backward(self, d, partial=None, prev_tail=None):
        """prev_tail: tail_info() of the last block of the previous group (None for group 1)."""
        for i in range(len(self.layers) - 1, -1, -1):
            pt = self.layers[i - 1].tail_info() if i > 0 else prev_tail
            d, partial = self.layers[i].backward(d, partial, pt)
        return d, partial


class Resnet(Layer):  # tf2/resnet.py:529-699
    def __init__(self, block_fn, layers, width_multiplier, cifar_stem=False, data_format='channels_last',
                 dropblock_keep_probs=None, dropblock_size=None, **kwargs):
        self.data_format = data_format
        if dropblock_keep_probs is None:
            dropblock_keep_probs = [None] * 4
        if not isinstance(dropblock_keep_probs, list) or len(dropblock_keep_probs) != 4:
            raise ValueError('dropblock_keep_probs is not valid:', dropblock_keep_probs)   # :546-547
        if FLAGS.train_mode == 'finetune' and FLAGS.fine_tune_after_block != -1:
            raise NotImplementedError('layer freezing (fine_tune_after_block) is outside the pretraining hot path')
        self.cifar_stem = cifar_stem
        self.resnet_d = (not cifar_stem) and FLAGS.sk_ratio > 0
        self.endpoints = {}
        with scope('resnet'):
            self.stem_pre = []      # ResNet-D: two extra (conv, BN+ReLU) pairs before the last stem conv
            if cifar_stem:                                                       # :551-564
                self.stem_conv = Conv2dFixedPadding(filters=64 * width_multiplier, kernel_size=3, strides=1,
                                                    data_format=data_format)
            elif self.resnet_d:                                                  # :566-591
                c0 = Conv2dFixedPadding(filters=64 * width_multiplier // 2, kernel_size=3, strides=2,
                                        data_format=data_format)
                b0 = BatchNormRelu(data_format=data_format)
                c1 = Conv2dFixedPadding(filters=64 * width_multiplier // 2, kernel_size=3, strides=1,
                                        data_format=data_format)
                b1 = BatchNormRelu(data_format=data_format)
                self.stem_pre = [c0, b0, c1, b1]
                self.stem_conv = Conv2dFixedPadding(filters=64 * width_multiplier, kernel_size=3, strides=1,
                                                    data_format=data_format)
            else:                                                                # :593-599
                self.stem_conv = Conv2dFixedPadding(filters=64 * width_multiplier, kernel_size=7, strides=2,
                                                    data_format=data_format)
            self.stem_bn = BatchNormRelu(data_format=data_format)                # :602-603
            self.block_groups = []
            for i, (f, s) in enumerate(zip([64, 128, 256, 512], [1, 2, 2, 2])):  # :620-668
                self.block_groups.append(BlockGroup(filters=f * width_multiplier, block_fn=block_fn,
                                                    blocks=layers[i], strides=s, name='block_group%d' % (i + 1),
                                                    data_format=data_format,
                                                    dropblock_keep_prob=dropblock_keep_probs[i],
                                                    dropblock_size=dropblock_size))
            self.block_groups[-1].layers[-1].is_final = True     # its output feeds the pooling: no consumer conv folds its tail

    @property
    def stem_kernel_stride(self):
        """(kernel, stride) of the conv that reads the 3-channel image."""
        if self.cifar_stem:
            return (3, 1)
        return (3, 2) if self.resnet_d else (7, 2)

    def __call__(self, inputs, training):
        """inputs: PackedInput (from Model) or an NHWC float32 tensor [B,H,W,3]."""
        if not isinstance(inputs, PackedInput):
            k, s = self.stem_kernel_stride
            inputs = PackedInput(inputs.contiguous(), 1, k, s, RT.dtype)
        if self.stem_pre:
            c0, b0, c1, b1 = self.stem_pre
            inputs = b1(c1(b0(c0(inputs, training), training), training), training)
        raw = self.stem_conv(inputs, training)
        self.endpoints['initial_conv'] = raw.t
        if self.cifar_stem:
            x = self.stem_bn(raw, training)
            self._pool = None
        else:                                                                    # BN+ReLU+maxpool, :602-611
            scale, shift = self.stem_bn.prepare(raw, training)
            y, arg = ops.bnrelu_maxpool_fwd(raw.t, scale, shift, 3, 2)
            self._pool = dict(arg=arg, H=raw.t.shape[1], W=raw.t.shape[2])
            x = Act(y)
        self.endpoints['initial_max_pool'] = x.t
        for i, g in enumerate(self.block_groups):
            x = g(x, training)
            self.endpoints['block_group%d' % (i + 1)] = x.t
        self._final = x.t
        # reduce_mean [1,2], :693-696.  With fp32 heads (--head_dtype=f32) the means leave the encoder in fp32.
        out = ops.global_avgpool_fwd(x.t, torch.float32 if FLAGS.head_dtype == 'f32' else None)
        self.endpoints['final_avg_pool'] = out
        return out

    def backward(self, dh, on_stage=None):
        """dh: [V, C] gradient wrt the pooled features.  on_stage(i) is called when block group
        i (4..1) has finished its backward, and on_stage(0) after the stem (gradient bucketing)."""
        _, H, W, _ = self._final.shape
        if dh.dtype != self._final.dtype:
            dh = ops.cast(dh, self._final.dtype)
        d = ops.global_avgpool_bwd(dh, H, W)
        self._final = None
        partial = None
        for i, g in reversed(list(enumerate(self.block_groups))):
            prev_tail = self.block_groups[i - 1].layers[-1].tail_info() if i > 0 else None
            d, partial = g.backward(d, partial, prev_tail)
            if on_stage is not None:
                on_stage(i + 1)
        if self._pool is not None:
            sb = self.stem_bn.saved
            C = sb['x'].shape[-1]
            epc = 16 // sb['x'].element_size()
            if _pool_fusion_enabled() and C % epc == 0 and C // epc <= 256 and 256 % (C // epc) == 0:
                # max-pool backward fused into the stem BN's backward reduce + apply: the un-pooled gradient
                # ([V,112,112,64], the largest tensor of the backward pass) is never written or re-read
                part = ops.bn_bwd_reduce_pool(d, self._pool['arg'], sb['x'], sb['scale'], sb['shift'], sb['mean'], sb['rstd'])
                c1, c2 = self.stem_bn._bwd_finalize(part, sb['count'])
                draw = ops.bn_bwd_apply_pool(d, self._pool['arg'], sb['x'], sb['scale'], sb['shift'], sb['mean'], sb['rstd'], c1, c2,
                                             ps_out=_ps_grad_ok(self.stem_conv))
                self.stem_bn.saved = None
            else:
                d = ops.maxpool_bwd(d, self._pool['arg'], self._pool['H'], self._pool['W'], 3, 2)
                # ReLU mask recomputed from x*scale+shift; the gradient goes to the stem's weight gradient only: pre-split where it takes it
                draw, _ = self.stem_bn.backward(d, mask_mode=2, ps_for=self.stem_conv)
            self._pool = None
        else:
            draw, _ = self.stem_bn.backward(d)
        if self.stem_pre:
            c0, b0, c1, b1 = self.stem_pre
            dm, part = self.stem_conv.backward(draw, fuse_bn=b1.fusion_info())
            dm, part = c1.backward(b1.backward_fused(dm, part), fuse_bn=b0.fusion_info())
            c0.backward(b0.backward_fused(dm, part), need_dx=False)
        else:
            self.stem_conv.backward(draw, need_dx=False)
        if on_stage is not None:
            on_stage(0)
        self.endpoints = {}
        return None


def resnet(resnet_depth, width_multiplier, cifar_stem=False, data_format='channels_last',
           dropblock_keep_probs=None, dropblock_size=None):
    """Returns the ResNet model for a given size (tf2/resnet.py:702-747)."""
    model_params = {
        18: {'block': ResidualBlock, 'layers': [2, 2, 2, 2]},
        34: {'block': ResidualBlock, 'layers': [3, 4, 6, 3]},
        50: {'block': BottleneckBlock, 'layers': [3, 4, 6, 3]},
        101: {'block': BottleneckBlock, 'layers': [3, 4, 23, 3]},
        152: {'block': BottleneckBlock, 'layers': [3, 8, 36, 3]},
        200: {'block': BottleneckBlock, 'layers': [3, 24, 36, 3]},
    }
    if resnet_depth not in model_params:
        raise ValueError('Not a valid resnet_depth:', resnet_depth)
    params = model_params[resnet_depth]
    return Resnet(params['block'], params['layers'], width_multiplier, cifar_stem=cifar_stem,
                  dropblock_keep_probs=dropblock_keep_probs, dropblock_size=dropblock_size,
                  data_format=data_format)
