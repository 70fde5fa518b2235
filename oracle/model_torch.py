"""Oracle: ResNet encoder + heads + pretraining step (torch-CPU, autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED TO THE REFERENCE'S SOURCE (tests/golden/reference_pin.npz: tf2/*.py executed on oracle/tfshim.py); TensorFlow's own kernels unpinned.

Restates /root/reference/tf2/resnet.py, /root/reference/tf2/model.py and the
step body /root/reference/tf2/run.py:557-622 with TensorFlow's semantics
(NHWC activations, HWIO conv kernels, [in,out] dense kernels, TF `SAME`
padding, Keras non-fused BatchNorm with biased variance).  Backward comes from
torch autograd, which plays the role of `tape.gradient` (run.py:621).

Weights live in an OrderedDict name -> torch tensor with Keras-style names so
the LARS name filters (tf2/model.py:36-42) behave as in the reference.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

BATCH_NORM_EPSILON = 1e-5  # tf2/resnet.py:28


@dataclass
class Config:
    """The subset of tf2/run.py flags the hot path reads (defaults = run.py:37-238)."""
    resnet_depth: int = 50
    width_multiplier: int = 1
    sk_ratio: float = 0.0
    image_size: int = 224
    num_classes: int = 1000
    proj_out_dim: int = 128
    num_proj_layers: int = 3
    proj_head_mode: str = 'nonlinear'
    ft_proj_selector: int = 0
    hidden_norm: bool = True
    temperature: float = 0.1
    batch_norm_decay: float = 0.9
    global_bn: bool = True
    weight_decay: float = 1e-6
    momentum: float = 0.9
    lineareval_while_pretraining: bool = True

    @property
    def cifar_stem(self):
        return self.image_size <= 32  # tf2/model.py:236


_MODEL_PARAMS = {  # tf2/resnet.py:709-734
    18: ('residual', [2, 2, 2, 2]), 34: ('residual', [3, 4, 6, 3]),
    50: ('bottleneck', [3, 4, 6, 3]), 101: ('bottleneck', [3, 4, 23, 3]),
    152: ('bottleneck', [3, 8, 36, 3]), 200: ('bottleneck', [3, 24, 36, 3]),
}


class _Namer:
    """Keras-style auto-numbered layer names ('conv2d', 'conv2d_1', ...)."""

    def __init__(self):
        self.c = {}

    def __call__(self, base):
        i = self.c.get(base, 0)
        self.c[base] = i + 1
        return base if i == 0 else '%s_%d' % (base, i)


def _trunc_normal(gen, shape, stddev):
    """VarianceScaling's truncated normal: resample outside 2 sigma."""
    out = torch.empty(shape, dtype=torch.float64)
    flat = out.view(-1)
    n = flat.numel()
    vals = torch.randn(n, generator=gen, dtype=torch.float64)
    bad = vals.abs() > 2
    while bad.any():
        vals[bad] = torch.randn(int(bad.sum()), generator=gen, dtype=torch.float64)
        bad = vals.abs() > 2
    flat.copy_(vals * stddev)
    return out


class Builder:
    """Walks the architecture once; used both to create parameters and to run it.

    mode 'init': creates parameters.  mode 'run': consumes them in the same
    order.  This keeps one definition of the layer sequence.
    """

    def __init__(self, cfg, params=None, state=None, seed=0, randomize_bn=False, dtype=torch.float32,
                 emulate_bf16=False):
        self.cfg = cfg
        # emulate_bf16: round weights and every stored activation to bfloat16 (straight-through in
        # backward) -- NOT reference behaviour; used only to calibrate how much drift a bf16-storage
        # implementation is expected to show against the exact restatement.
        self.q = (lambda t: t + (t.detach().bfloat16().to(t.dtype) - t.detach())) if emulate_bf16 else (lambda t: t)
        self.init = params is None
        self.params = OrderedDict() if params is None else params   # trainable
        self.state = OrderedDict() if state is None else state      # BN moving stats
        self.new_state = OrderedDict()
        self.namer = _Namer()
        self.gen = torch.Generator().manual_seed(seed)
        self.randomize_bn = randomize_bn
        self.dtype = dtype
        self.scope = []
        self.endpoints = OrderedDict()
        self.training = True

    # ---- variables -------------------------------------------------------
    def _name(self, leaf):
        return '/'.join(self.scope + [leaf])

    def _conv_kernel(self, layer, kh, kw, cin, cout):
        name = self._name(layer + '/kernel:0')
        if self.init:
            # tf.keras.initializers.VarianceScaling() (tf2/resnet.py:201):
            # scale=1, fan_in, truncated normal, stddev=sqrt(1/fan_in)/.87962566103423978
            std = math.sqrt(1.0 / (kh * kw * cin)) / .87962566103423978
            self.params[name] = _trunc_normal(self.gen, (kh, kw, cin, cout), std).to(self.dtype)
        return self.params[name]

    def _dense_kernel(self, layer, cin, cout):
        name = self._name(layer + '/kernel:0')
        if self.init:
            # RandomNormal(stddev=0.01) (tf2/model.py:145)
            self.params[name] = (torch.randn(cin, cout, generator=self.gen, dtype=torch.float64) * 0.01).to(self.dtype)
        return self.params[name]

    def _bias(self, layer, c):
        name = self._name(layer + '/bias:0')
        if self.init:
            self.params[name] = torch.zeros(c, dtype=self.dtype)
        return self.params[name]

    # ---- layers ----------------------------------------------------------
    def conv2d_fixed_padding(self, x, filters, kernel_size, strides):
        """Conv2dFixedPadding (tf2/resnet.py:183-208); x is NCHW here."""
        outer = self.namer('conv2d_fixed_padding')
        inner = self.namer('conv2d')
        cin = x.shape[1]
        w = self._conv_kernel('%s/%s' % (outer, inner), kernel_size, kernel_size, cin, filters)
        pad_total = kernel_size - 1
        beg = pad_total // 2
        end = pad_total - beg
        # strides>1: FixedPadding + VALID (:167-180,192-199); strides==1: SAME, which
        # for stride 1 pads (k-1)//2 before and the rest after -- the same numbers.
        x = F.pad(x, (beg, end, beg, end))
        return self.q(F.conv2d(x, self.q(w).permute(3, 2, 0, 1), stride=strides))

    def plain_conv1x1(self, x, filters):
        """Bare tf.keras.layers.Conv2D k=1 inside SK_Conv2D (tf2/resnet.py:243-256)."""
        inner = self.namer('conv2d')
        w = self._conv_kernel(inner, 1, 1, x.shape[1], filters)
        return F.conv2d(x, w.permute(3, 2, 0, 1))

    def batch_norm_relu(self, x, relu=True, init_zero=False, center=True, scale=True):
        """BatchNormRelu (tf2/resnet.py:31-78).  Training: batch mean and BIASED
        variance over every axis but channels (over all replicas if global_bn, :50-60);
        moving <- moving*decay + batch*(1-decay)."""
        outer = self.namer('batch_norm_relu')
        bn = self.namer('sync_batch_normalization' if self.cfg.global_bn else 'batch_normalization')
        base = self._name('%s/%s' % (outer, bn))
        c = x.shape[1]
        if self.init:
            if scale:
                g = torch.zeros(c) if init_zero else torch.ones(c)
                if self.randomize_bn:
                    g = 0.5 + torch.rand(c, generator=self.gen)
                self.params[base + '/gamma:0'] = g.to(self.dtype)
            if center:
                b = torch.zeros(c)
                if self.randomize_bn:
                    b = 0.1 * torch.randn(c, generator=self.gen)
                self.params[base + '/beta:0'] = b.to(self.dtype)
            self.state[base + '/moving_mean:0'] = torch.zeros(c, dtype=self.dtype)
            self.state[base + '/moving_variance:0'] = torch.ones(c, dtype=self.dtype)
        gamma = self.params[base + '/gamma:0'] if scale else None
        beta = self.params[base + '/beta:0'] if center else None
        mm = self.state[base + '/moving_mean:0']
        mv = self.state[base + '/moving_variance:0']
        axes = [0] + list(range(2, x.dim()))
        shape = [1, c] + [1] * (x.dim() - 2)
        if self.training:
            mean = x.mean(axes)
            var = ((x - mean.view(shape)) ** 2).mean(axes)
            d = self.cfg.batch_norm_decay
            self.new_state[base + '/moving_mean:0'] = (mm * d + mean.detach() * (1 - d))
            self.new_state[base + '/moving_variance:0'] = (mv * d + var.detach() * (1 - d))
        else:
            mean, var = mm, mv
        y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BATCH_NORM_EPSILON)
        if gamma is not None:
            y = y * gamma.view(shape)
        if beta is not None:
            y = y + beta.view(shape)
        self._last_bn_pre_relu = y
        return self.q(F.relu(y)) if relu else self.q(y)

    @staticmethod
    def _same_pad(size, k, s):
        out = -(-size // s)
        total = max((out - 1) * s + k - size, 0)
        return total // 2, total - total // 2

    def max_pool_same(self, x, k=3, s=2):
        """MaxPooling2D(pool 3, strides 2, 'SAME') (tf2/resnet.py:605-611)."""
        pt, pb = self._same_pad(x.shape[2], k, s)
        pl, pr = self._same_pad(x.shape[3], k, s)
        x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
        return F.max_pool2d(x, k, s)

    def avg_pool(self, x, strides):
        """AveragePooling2D(2, strides, SAME if strides==1 else VALID) preceded by
        FixedPadding(2) when strides>1 (tf2/resnet.py:330-338,400-408).  TF's SAME
        average excludes padded cells from the divisor."""
        if strides > 1:
            x = F.pad(x, (0, 1, 0, 1))   # FixedPadding(2): pad_total 1 -> (0,1)
            return F.avg_pool2d(x, 2, strides)
        pt, pb = self._same_pad(x.shape[2], 2, 1)
        pl, pr = self._same_pad(x.shape[3], 2, 1)
        ones = torch.ones_like(x[:1, :1])
        num = F.avg_pool2d(F.pad(x, (pl, pr, pt, pb)), 2, 1, divisor_override=1)
        den = F.avg_pool2d(F.pad(ones, (pl, pr, pt, pb)), 2, 1, divisor_override=1)
        return num / den

    def sk_conv2d(self, x, filters, strides):
        """SK_Conv2D (tf2/resnet.py:217-277)."""
        self.scope.append(self.namer('sk__conv2d'))
        x = self.conv2d_fixed_padding(x, 2 * filters, 3, strides)          # :264
        x = self.batch_norm_relu(x)                                          # :265
        streams = torch.stack(torch.split(x, filters, dim=1))                # :266  [2,B,f,H,W]
        g = streams.sum(0).mean((2, 3), keepdim=True)                        # :269-270
        mid = max(int(filters * self.cfg.sk_ratio), 32)                      # :242
        g = self.plain_conv1x1(g, mid)                                       # :271
        g = self.batch_norm_relu(g)                                          # :272
        mix = self.plain_conv1x1(g, 2 * filters)                             # :273
        mix = torch.stack(torch.split(mix, filters, dim=1))                  # :274
        mix = torch.softmax(mix, dim=0)                                      # :275
        self.scope.pop()
        return (streams * mix).sum(0)                                        # :277

    def _shortcut(self, x, filters_out, strides):
        if self.cfg.sk_ratio > 0:   # ResNet-D (:330-344, :400-414)
            x = self.avg_pool(x, strides)
            x = self.conv2d_fixed_padding(x, filters_out, 1, 1)
        else:
            x = self.conv2d_fixed_padding(x, filters_out, 1, strides)
        return self.batch_norm_relu(x, relu=False)

    def residual_block(self, x, filters, strides, use_projection):
        """ResidualBlock (tf2/resnet.py:314-382)."""
        self.scope.append(self.namer('residual_block'))
        shortcut = self._shortcut(x, filters, strides) if use_projection else x
        x = self.conv2d_fixed_padding(x, filters, 3, strides)
        x = self.batch_norm_relu(x)
        x = self.conv2d_fixed_padding(x, filters, 3, 1)
        x = self.batch_norm_relu(x, relu=False, init_zero=True)
        self.scope.pop()
        return self.q(F.relu(x + shortcut))

    def bottleneck_block(self, x, filters, strides, use_projection):
        """BottleneckBlock (tf2/resnet.py:385-487); DropBlock layers are identity."""
        self.scope.append(self.namer('bottleneck_block'))
        shortcut = self._shortcut(x, 4 * filters, strides) if use_projection else x
        x = self.conv2d_fixed_padding(x, filters, 1, 1)                      # :431-435
        x = self.batch_norm_relu(x)
        if self.cfg.sk_ratio > 0:
            x = self.sk_conv2d(x, filters, strides)                          # :442-444
        else:
            x = self.conv2d_fixed_padding(x, filters, 3, strides)            # :446-453
            x = self.batch_norm_relu(x)
        x = self.conv2d_fixed_padding(x, 4 * filters, 1, 1)                  # :460-467
        x = self.batch_norm_relu(x, relu=False, init_zero=True)
        self.scope.pop()
        return self.q(F.relu(x + shortcut))                                  # :487

    def resnet(self, x):
        """Resnet.call (tf2/resnet.py:683-699); x NHWC in, [B, C] out."""
        cfg = self.cfg
        kind, layers = _MODEL_PARAMS[cfg.resnet_depth]
        w = cfg.width_multiplier
        self.scope.append('resnet')
        x = x.permute(0, 3, 1, 2)
        if cfg.cifar_stem:                                                   # :551-564
            x = self.conv2d_fixed_padding(x, 64 * w, 3, 1)
            self.endpoints['initial_conv'] = x
            x = self.batch_norm_relu(x)
            self.endpoints['initial_max_pool'] = x
        else:
            if cfg.sk_ratio > 0:                                             # :566-591
                x = self.conv2d_fixed_padding(x, 64 * w // 2, 3, 2)
                x = self.batch_norm_relu(x)
                x = self.conv2d_fixed_padding(x, 64 * w // 2, 3, 1)
                x = self.batch_norm_relu(x)
                x = self.conv2d_fixed_padding(x, 64 * w, 3, 1)
            else:                                                            # :593-599
                x = self.conv2d_fixed_padding(x, 64 * w, 7, 2)
            self.endpoints['initial_conv'] = x
            x = self.batch_norm_relu(x)                                      # :602-603
            x = self.max_pool_same(x)                                        # :605-611
            self.endpoints['initial_max_pool'] = x
        block = self.residual_block if kind == 'residual' else self.bottleneck_block
        for gi, (f, nb, s) in enumerate(zip([64, 128, 256, 512], layers, [1, 2, 2, 2])):
            self.scope.append('block_group%d' % (gi + 1))
            x = block(x, f * w, s, True)                                     # :498-506
            for _ in range(1, nb):
                x = block(x, f * w, 1, False)                                # :508-515
            self.scope.pop()
            self.endpoints['block_group%d' % (gi + 1)] = x
        x = self.q(x.mean((2, 3)))                                           # :693-696
        self.endpoints['final_avg_pool'] = x
        self.scope.pop()
        return x

    def linear_layer(self, x, num_classes, use_bias=True, use_bn=False, name='linear_layer'):
        """LinearLayer (tf2/model.py:119-154)."""
        self.scope.append(name)
        dense = self.namer('dense')
        w = self._dense_kernel(dense, x.shape[1], num_classes)
        y = self.q(x @ self.q(w))
        if use_bias and not use_bn:                                          # :146
            y = y + self._bias(dense, num_classes)
        if use_bn:
            y = self.batch_norm_relu(y, relu=False, center=use_bias)         # :134-135,152-153
        self.scope.pop()
        return y

    def projection_head(self, h):
        """ProjectionHead.call (tf2/model.py:192-213), mode 'nonlinear'/'none'."""
        cfg = self.cfg
        if cfg.proj_head_mode == 'none':
            return h, h
        assert cfg.proj_head_mode == 'nonlinear'
        self.scope.append('projection_head')
        hiddens = [h]
        for j in range(cfg.num_proj_layers):
            if j != cfg.num_proj_layers - 1:
                y = self.linear_layer(hiddens[-1], hiddens[-1].shape[1], True, True, 'nl_%d' % j)
                y = self.q(F.relu(y))                                        # :204-205
            else:
                y = self.linear_layer(hiddens[-1], cfg.proj_out_dim, False, True, 'nl_%d' % j)
            hiddens.append(y)
        self.scope.pop()
        return hiddens[-1], hiddens[cfg.ft_proj_selector]

    def model(self, inputs, training=True):
        """Model.__call__ (tf2/model.py:241-280), train_mode='pretrain', use_blur=False."""
        self.training = training
        k = inputs.shape[3] // 3
        feats = self.q(torch.cat(torch.split(inputs, 3, dim=3), 0))          # :250-259
        self.scope.append('model')
        h = self.resnet(feats)                                               # :262
        proj, sup_in = self.projection_head(h)                               # :265-266
        sup = None
        if self.cfg.lineareval_while_pretraining:                            # :272-278
            self.scope.append('head_supervised')
            sup = self.linear_layer(sup_in.detach(), self.cfg.num_classes)
            self.scope.pop()
        self.scope.pop()
        return proj, sup


def init_model(cfg, seed=0, randomize_bn=False, image_shape=None, dtype=torch.float32):
    """Create (params, state) by tracing a 1-image forward pass."""
    b = Builder(cfg, seed=seed, randomize_bn=randomize_bn, dtype=dtype)
    hw = cfg.image_size if image_shape is None else image_shape
    with torch.no_grad():
        b.model(torch.zeros(2, hw, hw, 6, dtype=dtype), training=True)
    return b.params, b.state


def torch_contrastive_loss(hidden, hidden_norm, temperature):
    """tf2/objective.py:35-89, strategy=None branch, differentiable torch."""
    if hidden_norm:
        hidden = hidden * torch.rsqrt(torch.clamp((hidden * hidden).sum(-1, keepdim=True), min=1e-12))
    n = hidden.shape[0] // 2
    h1, h2 = hidden[:n], hidden[n:]
    masks = torch.eye(n, dtype=hidden.dtype)
    labels = torch.cat([torch.eye(n, dtype=hidden.dtype), torch.zeros(n, n, dtype=hidden.dtype)], 1)
    laa = h1 @ h1.T / temperature - masks * 1e9
    lbb = h2 @ h2.T / temperature - masks * 1e9
    lab = h1 @ h2.T / temperature
    lba = h2 @ h1.T / temperature
    la = torch.cat([lab, laa], 1)
    lb = torch.cat([lba, lbb], 1)
    loss_a = -(labels * torch.log_softmax(la, 1)).sum(1)
    loss_b = -(labels * torch.log_softmax(lb, 1)).sum(1)
    return (loss_a + loss_b).mean(), lab, labels, hidden


def single_step_losses(cfg, params, state, images, labels_onehot, emulate_bf16=False):
    """Loss composition of tf2/run.py:577-617 for one replica (R=1).

    Returns dict with total loss tensor (differentiable), pieces, new BN state,
    the normalised embeddings and the endpoints.
    """
    b = Builder(cfg, params=params, state=state, emulate_bf16=emulate_bf16)
    proj, sup = b.model(images, training=True)
    con_loss, logits_con, labels_con, z = torch_contrastive_loss(proj, cfg.hidden_norm, cfg.temperature)
    loss = con_loss
    out = {'con_loss': con_loss, 'logits_con': logits_con, 'labels_con': labels_con,
           'z': z, 'proj': proj, 'endpoints': b.endpoints}
    if sup is not None:
        l = torch.cat([labels_onehot, labels_onehot], 0)                     # run.py:599-600
        sup_loss = -(l * torch.log_softmax(sup, 1)).sum(1).mean()            # objective.py:27-32
        out['sup_loss'] = sup_loss
        out['sup_logits'] = sup
        loss = loss + sup_loss
    # add_weight_decay, LARS branch (tf2/model.py:49-60): sup-head kernel only
    wd = 0.0
    for name, p in params.items():
        if 'head_supervised' in name and 'bias' not in name:
            wd = wd + 0.5 * (p * p).sum()
    wd = cfg.weight_decay * wd
    out['weight_decay'] = wd
    out['total_loss'] = loss + wd                                            # run.py:612-613
    out['new_state'] = b.new_state
    return out


def train_step(cfg, params, state, momenta, images, labels_onehot, learning_rate, emulate_bf16=False):
    """One full tf2/run.py:557-622 step on CPU (R=1): forward, autograd backward,
    LARS (tf2/lars_optimizer.py:83-137 via torch ops, fp32).  Mutates nothing;
    returns (new_params, new_state, new_momenta, info)."""
    ps = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    out = single_step_losses(cfg, ps, state, images, labels_onehot, emulate_bf16=emulate_bf16)
    total = out['total_loss']                                                # / R with R = 1
    grads = torch.autograd.grad(total, list(ps.values()), allow_unused=True)
    exclude = ['batch_normalization', 'bias', 'head_supervised']             # tf2/model.py:39-41
    import re
    new_p, new_m = OrderedDict(), OrderedDict()
    with torch.no_grad():
        for (name, p), g in zip(ps.items(), grads):
            if g is None:
                g = torch.zeros_like(p)
            v = momenta[name]
            excluded = any(re.search(r, name) for r in exclude)
            if cfg.weight_decay and not excluded:
                g = g + cfg.weight_decay * p
            trust = 1.0
            if not excluded:
                wn, gn = p.norm(), g.norm()
                if wn > 0 and gn > 0:
                    trust = 0.001 * wn / gn
            slr = learning_rate * trust
            nv = cfg.momentum * v + slr * g
            new_m[name] = nv
            new_p[name] = p.detach() - nv
    new_state = OrderedDict(state)
    new_state.update(out['new_state'])
    out['grads'] = OrderedDict(zip(ps.keys(), grads))
    return new_p, new_state, new_m, out
