"""Model assembly -- drop-in mirror of /root/reference/tf2/model.py on the HIP kernels.

Mirrors `build_optimizer` (:29-44), `add_weight_decay` (:47-69), `get_train_steps` (:72-75),
`WarmUpAndCosineDecay` (:78-116), `LinearLayer` (:119-154), `ProjectionHead` (:157-213),
`SupervisedHead` (:216-225) and `Model` (:228-280) with the reference's names, argument
order and defaults.  Forward results are plain device tensors; each class also carries the
hand-written backward that `tape.gradient` (tf2/run.py:621) would derive.
"""
import math

import torch

from . import data_util, lars_optimizer, ops, optimizers, resnet
from .flags import FLAGS
from .lars_optimizer import Variable
from .resnet import RT, Act, Layer, PackedInput, scope


def build_optimizer(learning_rate):
    """Returns the optimizer (tf2/model.py:29-44)."""
    if FLAGS.optimizer == 'lars':
        return lars_optimizer.LARSOptimizer(
            learning_rate,
            momentum=FLAGS.momentum,
            weight_decay=FLAGS.weight_decay,
            exclude_from_weight_decay=['batch_normalization', 'bias', 'head_supervised'])
    elif FLAGS.optimizer == 'momentum':                                                     # :31-32
        # l2: the gradient of add_weight_decay's loss term for the non-LARS optimizers (:62-69), applied inside the update kernel
        return optimizers.SGD(learning_rate, FLAGS.momentum, nesterov=True, l2=FLAGS.weight_decay)
    elif FLAGS.optimizer == 'adam':                                                         # :33-34
        return optimizers.Adam(learning_rate, l2=FLAGS.weight_decay)
    else:
        raise ValueError('Unknown optimizer {}'.format(FLAGS.optimizer))


def add_weight_decay(model, adjust_per_optimizer=True):
    """Compute weight decay from flags (tf2/model.py:47-69).  Returns a float32 device scalar
    (or 0).  With LARS only the supervised head's non-bias variables contribute (:49-60); the
    matching gradient term wd*w is added by Model.backward."""
    if adjust_per_optimizer and 'lars' in FLAGS.optimizer:
        vs = [v for v in model.trainable_variables if 'head_supervised' in v.name and 'bias' not in v.name]
    else:
        vs = [v for v in model.trainable_variables if 'batch_normalization' not in v.name]
    if not vs:
        return 0
    nv = len(vs)
    out = ops.step_scalars(nv + 1, vs[0].value.device)
    for i, v in enumerate(vs):
        ops.l2_loss_f32(v.value, out[i:i + 1])          # tf.nn.l2_loss = sum(v^2)/2
    if nv > 16:                                         # SGD / Adam: every non-BN variable contributes
        return FLAGS.weight_decay * out[:nv].sum()
    # weight_decay * add_n(l2 losses) (:58-60 / :66-68) in one launch
    ops.accumulate_scalars([out[j:j + 1] for j in range(nv)], scales=[FLAGS.weight_decay] * nv, total=out[nv:nv + 1],
                           total_mask=(1 << nv) - 1)
    return out[nv]


def get_train_steps(num_examples):
    """Determine the number of training steps (tf2/model.py:72-75)."""
    return FLAGS.train_steps or (num_examples * FLAGS.train_epochs // FLAGS.train_batch_size + 1)


class WarmUpAndCosineDecay:
    """Applies a warmup schedule on a given learning rate decay schedule (tf2/model.py:78-116)."""

    def __init__(self, base_learning_rate, num_examples, name=None):
        self.base_learning_rate = base_learning_rate
        self.num_examples = num_examples
        self._name = name

    def __call__(self, step):
        warmup_steps = int(round(FLAGS.warmup_epochs * self.num_examples // FLAGS.train_batch_size))   # :89-91
        if FLAGS.learning_rate_scaling == 'linear':
            scaled_lr = self.base_learning_rate * FLAGS.train_batch_size / 256.
        elif FLAGS.learning_rate_scaling == 'sqrt':
            scaled_lr = self.base_learning_rate * math.sqrt(FLAGS.train_batch_size)
        else:
            raise ValueError('Unknown learning rate scaling {}'.format(FLAGS.learning_rate_scaling))
        learning_rate = (step / float(warmup_steps) * scaled_lr if warmup_steps else scaled_lr)
        total_steps = get_train_steps(self.num_examples)
        decay_steps = total_steps - warmup_steps
        # tf.keras.experimental.CosineDecay(scaled_lr, decay_steps)(step - warmup_steps), alpha=0
        # decay_steps <= 0 (train_steps <= warmup_steps): the reference divides 0/0 here; the decay is then complete
        s = min(max(step - warmup_steps, 0), decay_steps)
        frac = (s / decay_steps) if decay_steps > 0 else 1.0
        cosine = scaled_lr * 0.5 * (1.0 + math.cos(math.pi * frac))
        return learning_rate if step < warmup_steps else cosine                       # :107-108

    def get_config(self):
        return {'base_learning_rate': self.base_learning_rate, 'num_examples': self.num_examples}


class LinearLayer(Layer):  # tf2/model.py:119-154
    def __init__(self, num_classes, use_bias=True, use_bn=False, name='linear_layer', **kwargs):
        # Note: use_bias is ignored for the dense layer when use_bn=True (it is used for BN's center).
        self.num_classes = num_classes
        self.use_bias = use_bias
        self.use_bn = use_bn
        self._name = name
        with scope(name):
            if self.use_bn:
                self.bn_relu = resnet.BatchNormRelu(relu=False, center=use_bias)      # :134-135
            self._dense = RT.unique('dense')
            self._path = RT.path(self._dense)
        self.kernel = None
        self.bias = None
        self._version = -1
        self.saved = None

    def build(self, cin):
        n = self.num_classes(cin) if callable(self.num_classes) else self.num_classes
        self.nout = n
        self.npad = (n + 15) // 16 * 16
        RT.seed += 1
        g = torch.Generator().manual_seed(RT.seed)
        w = torch.randn(cin, n, generator=g) * 0.01                   # RandomNormal(stddev=.01), :145
        self.kernel = Variable(self._path + '/kernel:0', w.to(RT.device))
        if self.use_bias and not self.use_bn:                        # :146
            self.bias = Variable(self._path + '/bias:0', torch.zeros(n, device=RT.device))
        self.cin = cin

    def _refresh(self, dtype):
        """Compute copies of the fp32 master weight in the dtype of the layer's input."""
        if self._version == RT.weights_version and getattr(self, '_dtype', None) == dtype:
            return
        w4 = self.kernel.value.view(1, 1, self.cin, self.nout)
        if self.npad == self.nout:
            self.w_t = ops.prep_weights(w4, 0, dtype)
        else:   # class dimension padded to a multiple of 16 with zero rows
            self.w_t = torch.zeros(self.npad, self.cin, device=RT.device, dtype=dtype)
            ops.prep_weights(w4, 0, dtype, out=self.w_t[:self.nout])
        self.w_d = ops.prep_weights(w4, 1, dtype) if self.npad == self.nout else None
        self._version = RT.weights_version
        self._dtype = dtype

    def __call__(self, inputs, training, relu=False):
        """inputs: [V, C] tensor.  Returns [V, n] tensor (dense [+BN [+relu]])."""
        assert inputs.dim() == 2, inputs.shape
        V, cin = inputs.shape
        if self.kernel is None:
            self.build(cin)
        self._refresh(inputs.dtype)
        x4 = inputs.view(V, 1, 1, cin)
        stats = ops.conv_stats(V, self.npad, RT.device) if (self.use_bn and training) else None
        y, stats, sums = ops.conv2d_fwd_with_stats(x4, self.w_t, 1, 1, 1, 0, 1, 1, stats)
        y = y.view(V, self.npad)
        self.saved = dict(x=x4)
        if self.use_bn:
            return self.bn_relu(Act(y, stats, sums=sums), training, relu=relu).t
        return y

    def backward(self, dy, need_dx=True):
        if self.use_bn:
            dy, _ = self.bn_relu.backward(dy)
        V = dy.shape[0]
        dy4 = dy.view(V, 1, 1, self.npad)
        x4 = self.saved['x']
        self.saved = None
        g = self.kernel.ensure_grad()
        if self.npad == self.nout:
            ops.conv2d_wgrad(x4, dy4, 1, 1, 1, 0, out=g)
        else:
            tmp = ops.conv2d_wgrad(x4, dy4, 1, 1, 1, 0)
            g.copy_(tmp[:, :self.nout])
        if self.bias is not None:
            ops.colsum(dy, self.nout, self.bias.ensure_grad())
        if not need_dx:
            return None
        return ops.conv2d_dgrad(dy4, self.w_d, 1, 1, 1, 0, 1, 1).view(V, self.cin)


class ProjectionHead(Layer):  # tf2/model.py:157-213
    def __init__(self, **kwargs):
        out_dim = FLAGS.proj_out_dim
        self.linear_layers = []
        with scope('projection_head'):
            if FLAGS.proj_head_mode == 'none':
                pass  # directly use the output hiddens as hiddens
            elif FLAGS.proj_head_mode == 'linear':
                self.linear_layers = [LinearLayer(num_classes=out_dim, use_bias=False, use_bn=True, name='l_0')]
            elif FLAGS.proj_head_mode == 'nonlinear':
                for j in range(FLAGS.num_proj_layers):
                    if j != FLAGS.num_proj_layers - 1:
                        # for the middle layers, use bias and relu for the output.
                        self.linear_layers.append(LinearLayer(num_classes=lambda cin: int(cin), use_bias=True,
                                                              use_bn=True, name='nl_%d' % j))
                    else:
                        # for the final layer, neither bias nor relu is used.
                        self.linear_layers.append(LinearLayer(num_classes=FLAGS.proj_out_dim, use_bias=False,
                                                              use_bn=True, name='nl_%d' % j))
            else:
                raise ValueError('Unknown head projection mode {}'.format(FLAGS.proj_head_mode))

    def __call__(self, inputs, training):
        if FLAGS.proj_head_mode == 'none':
            return inputs, inputs  # directly use the output hiddens as hiddens
        self._in_dtype = inputs.dtype
        if FLAGS.head_dtype == 'f32' and inputs.dtype != torch.float32:
            inputs = ops.cast(inputs, torch.float32)          # heads in fp32 on top of a bf16 encoder
        hiddens_list = [inputs]
        if FLAGS.proj_head_mode == 'linear':
            # The reference returns None here (list.append, tf2/model.py:198-199); we return the
            # evidently intended pair instead of reproducing the bug.
            hiddens_list.append(self.linear_layers[0](hiddens_list[-1], training))
        else:
            n = FLAGS.num_proj_layers
            for j in range(n):
                hiddens_list.append(self.linear_layers[j](hiddens_list[-1], training, relu=(j != n - 1)))
        # The first element is the output of the projection head, the second the finetune-head input.
        return hiddens_list[-1], hiddens_list[FLAGS.ft_proj_selector]

    def backward(self, d):
        for layer in reversed(self.linear_layers):
            d = layer.backward(d)
        if self.linear_layers and d.dtype != self._in_dtype:
            d = ops.cast(d, self._in_dtype)
        return d


class SupLogits:
    """Supervised-head output before the bias add; bias + softmax-CE + gradient are one fused
    launch in objective.add_supervised_loss."""

    def __init__(self, z, bias, num_classes):
        self.z, self.bias, self.num_classes = z, bias, num_classes

    def dense(self):
        return self.z[:, :self.num_classes].float() + self.bias


class SupervisedHead(Layer):  # tf2/model.py:216-225
    def __init__(self, num_classes, name='head_supervised', **kwargs):
        with scope(name):
            self.linear_layer = LinearLayer(num_classes)

    def __call__(self, inputs, training):
        z = self.linear_layer(inputs, training)
        return SupLogits(z, self.linear_layer.bias.value, self.linear_layer.nout)

    def backward(self, dlogits):
        self.linear_layer.backward(dlogits, need_dx=False)      # stop_gradient on the input, :276-277


class Model(Layer):
    """Resnet model with projection or supervised layer (tf2/model.py:228-280)."""

    def __init__(self, num_classes, **kwargs):
        RT.strategy = kwargs.get('strategy', RT.strategy)
        with scope('model'):
            self.resnet_model = resnet.resnet(resnet_depth=FLAGS.resnet_depth,
                                              width_multiplier=FLAGS.width_multiplier,
                                              cifar_stem=FLAGS.image_size <= 32)
            self._projection_head = ProjectionHead()
            self.supervised_head = None
            if FLAGS.train_mode == 'finetune' or FLAGS.lineareval_while_pretraining:
                self.supervised_head = SupervisedHead(num_classes)
        self._flat_grads = None

    def __call__(self, inputs, training):
        """inputs: float32 [b, H, W, 3k] in [0,1].  Returns (projection_head_outputs float32
        [k*b, proj_out_dim], supervised_head_outputs SupLogits) like tf2/model.py:241-280."""
        if training and FLAGS.train_mode == 'pretrain':
            if FLAGS.fine_tune_after_block > -1:
                raise ValueError('Does not support layer freezing during pretraining,'
                                 'should set fine_tune_after_block<=-1 for safety.')
        if inputs.dim() != 4 or inputs.shape[3] % 3 != 0:
            raise ValueError('The input channels dimension must be statically known '
                             f'(got input shape {tuple(inputs.shape)})')
        if FLAGS.train_mode == 'finetune':
            raise NotImplementedError('train_mode=finetune is outside the pretraining hot path')
        # (evaluation passes do not go through ops.begin_step.)  Inference mode normalises with the MOVING statistics, so nothing bounds the
        # activations to fp16's range (a freshly initialised network's moving averages do not normalise at all): the split-fp16 forward is
        # for training-mode BatchNorm only, an inference forward runs its fp32 products as six bf16 terms instead (same accuracy class).
        ops.select_f32_matmul(inference=not training)
        if FLAGS.use_blur and training and FLAGS.train_mode == 'pretrain':
            # batch_random_blur on the device (tf2/model.py:255-258), fused over the k views
            inputs = data_util.batch_random_blur_tensor(inputs, FLAGS.image_size, FLAGS.image_size)
        num_transforms = inputs.shape[3] // 3
        k, s = self.resnet_model.stem_kernel_stride
        # (training: the 7x7 stem's weight gradient reads the image as bf16 pieces in the three-term modes -- written by the packing pass)
        rm = self.resnet_model
        ps_cout = rm.stem_conv.filters if (training and not rm.stem_pre and not rm.cifar_stem) else 0
        packed = PackedInput(inputs.contiguous(), num_transforms, k, s, RT.dtype, presplit_for_cout=ps_cout)   # split + concat, :250-259
        hiddens = self.resnet_model(packed, training=training)                     # :262
        proj, sup_in = self._projection_head(hiddens, training)                    # :265-266
        self._proj_is_encoder = proj is hiddens
        self._proj_dtype = proj.dtype
        proj32 = ops.cast(proj, torch.float32) if proj.dtype != torch.float32 else proj
        sup_out = None
        if FLAGS.train_mode == 'pretrain' and FLAGS.lineareval_while_pretraining:
            sup_out = self.supervised_head(sup_in, training)                       # stop_gradient, :276-278
        return proj32, sup_out

    def backward_supervised(self, d_sup):
        """Backward of the linear-eval head alone (its input is stop_gradient'ed, tf2/model.py:276-277)."""
        if d_sup is not None:
            self.supervised_head.backward(d_sup)
            if 'lars' in FLAGS.optimizer and FLAGS.weight_decay:   # d/dw of add_weight_decay, :49-60
                k = self.supervised_head.linear_layer.kernel
                ops.axpy_f32(FLAGS.weight_decay * self._wd_grad_scale, k.value, k.grad)

    def backward(self, d_proj, d_sup=None, on_stage=None):
        """d_proj: float32 [k*b, proj_out_dim]; d_sup: gradient wrt the supervised logits."""
        self.backward_supervised(d_sup)
        d = ops.cast(d_proj, self._proj_dtype) if self._proj_dtype != torch.float32 else d_proj
        d = self._projection_head.backward(d)
        self.resnet_model.backward(d, on_stage=on_stage)

    _wd_grad_scale = 1.0   # set to 1/num_replicas by the step (loss / R, tf2/run.py:617)

    def allocate_flat_grads(self):
        """One flat fp32 buffer for every trainable gradient (ordered last-layer-first so the
        gradient all-reduce can be bucketed along the backward pass)."""
        vs = list(reversed(self.trainable_variables))
        offs, total = [], 0
        for v in vs:
            offs.append(total)
            total += (v.numel() + 63) // 64 * 64
        flat = torch.zeros(total, device=vs[0].value.device, dtype=torch.float32)
        for v, o in zip(vs, offs):
            v.grad = flat[o:o + v.numel()].view(v.value.shape)
        self._flat_grads = flat
        self._flat_order = vs
        self._flat_offsets = offs
        return flat
