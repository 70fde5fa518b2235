"""The two non-LARS branches of `build_optimizer` (/root/reference/tf2/model.py:29-44):
`tf.keras.optimizers.SGD(learning_rate, FLAGS.momentum, nesterov=True)` and `tf.keras.optimizers.Adam(learning_rate)`.

Same constructor arguments and defaults as the Keras classes the reference instantiates, same object surface as
`lars_optimizer.LARSOptimizer` (`iterations`, `apply_gradients`, `get_slot`, the `_slots` the checkpoint saves); the update
is ONE multi-tensor launch over every trainable tensor on the LARS descriptor / chunk tables
(simclr_sgd_multi_tensor / simclr_adam_multi_tensor, csrc/lars.hip).

Weight decay: with these optimizers the reference adds `weight_decay * sum l2_loss(v)` over the non-BatchNorm variables to the
LOSS (tf2/model.py:62-69); its gradient, `weight_decay * v`, is added inside the update kernel (`l2`, per-tensor flag) instead
of by a separate pass over the weights -- after the gradient all-reduce, so the full coefficient, not 1/R of it.
"""
import ctypes

import torch

from ._lib import lib


class _MultiTensorOptimizer:
    slot_factor = 1          # floats of slot state per parameter element

    def __init__(self, learning_rate, l2=0.0, l2_exclude=('batch_normalization',), name=None):
        self.name = name
        self.learning_rate = learning_rate          # float or callable(step) (a schedule)
        self.l2 = float(l2 or 0.0)
        self.l2_exclude = tuple(l2_exclude or ())
        self.iterations = 0
        self._slots = {}
        self._key = None
        self._table = self._chunks = None

    def _create_slots(self, var_list):
        for v in var_list:
            if id(v) not in self._slots:
                shape = ((self.slot_factor,) if self.slot_factor > 1 else ()) + tuple(v.value.shape)
                self._slots[id(v)] = torch.zeros(shape, device=v.value.device, dtype=torch.float32)

    def _takes_l2(self, name):
        return bool(self.l2) and not any(s in name for s in self.l2_exclude)

    def _build(self, variables, grads):
        self._create_slots(variables)
        T = len(variables)
        chunk = lib().lars_chunk_elems()
        table = torch.zeros(5 * T, dtype=torch.int64)
        chunks = []
        for t, (v, g) in enumerate(zip(variables, grads)):
            assert v.value.dtype == torch.float32 and g.dtype == torch.float32
            assert v.value.is_contiguous() and g.is_contiguous() and g.shape == v.value.shape
            table[0 * T + t] = v.value.data_ptr()
            table[1 * T + t] = g.data_ptr()
            table[2 * T + t] = self._slots[id(v)].data_ptr()
            table[3 * T + t] = v.value.numel()
            table[4 * T + t] = 1 if self._takes_l2(v.name) else 0
            for off in range(0, v.value.numel(), chunk):
                chunks.append((t, off))
        dev = variables[0].value.device
        self._table = table.to(dev)
        self._chunks = torch.tensor(chunks, dtype=torch.int64).view(-1).to(dev)
        self._num = (T, len(chunks))
        self._key = tuple((v.value.data_ptr(), g.data_ptr()) for v, g in zip(variables, grads))

    def current_lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def apply_gradients(self, grads_and_vars, lr_device=None):
        """tf2/run.py:622.  grads_and_vars: iterable of (grad tensor, Variable)."""
        pairs = [(g, v) for g, v in grads_and_vars if g is not None and v is not None]
        grads = [g for g, _ in pairs]
        variables = [v for _, v in pairs]
        key = tuple((v.value.data_ptr(), g.data_ptr()) for v, g in zip(variables, grads))
        if key != self._key:
            self._build(variables, grads)
        T, nchunks = self._num
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lrp = ctypes.c_void_p(lr_device.data_ptr()) if lr_device is not None else None
        self._launch(ctypes.c_void_p(self._table.data_ptr()), T, ctypes.c_void_p(self._chunks.data_ptr()), nchunks, lrp,
                     self.current_lr(), stream)
        self.iterations += 1


class SGD(_MultiTensorOptimizer):
    """tf.keras.optimizers.SGD(learning_rate=0.01, momentum=0.0, nesterov=False) -- the reference passes
    (learning_rate, FLAGS.momentum, nesterov=True), tf2/model.py:31-32.  accum = momentum * accum - lr * g;
    w += momentum * accum - lr * g (nesterov) | accum."""

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, name='SGD', l2=0.0, l2_exclude=('batch_normalization',)):
        super().__init__(learning_rate, l2, l2_exclude, name)
        if not 0.0 <= float(momentum) <= 1.0:
            raise ValueError('`momentum` must be between [0, 1].')          # Keras' own check
        self.momentum = float(momentum)
        self.nesterov = bool(nesterov)

    def get_slot(self, var, slot_name='momentum'):
        assert slot_name == 'momentum'
        return self._slots[id(var)]

    def _launch(self, table, T, chunks, nchunks, lrp, lr, stream):
        lib().sgd_multi_tensor(table, T, chunks, nchunks, lrp, lr, self.momentum, int(self.nesterov), self.l2, stream)

    def get_config(self):
        return {'name': self.name, 'learning_rate': self.learning_rate if not callable(self.learning_rate) else 'schedule',
                'momentum': self.momentum, 'nesterov': self.nesterov}


class Adam(_MultiTensorOptimizer):
    """tf.keras.optimizers.Adam(learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False), tf2/model.py:33-34:
    m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; w -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + epsilon)."""
    slot_factor = 2          # the slot of a variable is [2, *shape]: m, v

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, name='Adam', l2=0.0,
                 l2_exclude=('batch_normalization',)):
        super().__init__(learning_rate, l2, l2_exclude, name)
        if amsgrad:
            raise NotImplementedError('amsgrad=True is not used by the reference (tf2/model.py:34)')
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)

    def get_slot(self, var, slot_name):
        assert slot_name in ('m', 'v')
        return self._slots[id(var)][0 if slot_name == 'm' else 1]

    def _launch(self, table, T, chunks, nchunks, lrp, lr, stream):
        lib().adam_multi_tensor(table, T, chunks, nchunks, lrp, lr, self.beta_1, self.beta_2, self.epsilon, self.iterations + 1,
                                self.l2, stream)

    def get_config(self):
        return {'name': self.name, 'learning_rate': self.learning_rate if not callable(self.learning_rate) else 'schedule',
                'beta_1': self.beta_1, 'beta_2': self.beta_2, 'epsilon': self.epsilon, 'amsgrad': False}
