"""LARS optimizer -- drop-in mirror of /root/reference/tf2/lars_optimizer.py.

Same constructor signature, defaults and name-regex semantics as
`LARSOptimizer` (tf2/lars_optimizer.py:25-77, :139-157); the update itself
(`_resource_apply_dense`, :83-137) runs as ONE fused multi-tensor HIP launch pair
over every trainable tensor (simclr_lars_multi_tensor in include/simclr_hip.h)
instead of one TF op group per variable.
"""
import ctypes
import re

import torch

from ._lib import lib

EETA_DEFAULT = 0.001  # tf2/lars_optimizer.py:22


class Variable:
    """A named trainable tensor (the stand-in for a tf.Variable).

    `value` is the fp32 master copy (device tensor); `grad` is a persistent fp32
    gradient buffer of the same shape.  `name` keeps the Keras-style spelling because
    LARS filters on it (tf2/model.py:36-42).
    """

    def __init__(self, name, value, trainable=True):
        self.name = name
        self.value = value
        self.grad = None
        self.trainable = trainable

    @property
    def shape(self):
        return tuple(self.value.shape)

    def numel(self):
        return self.value.numel()

    def ensure_grad(self):
        if self.grad is None:
            # slot rounded up to 64 elements: kernels that work on channel-padded tensors may write
            # (zeros) up to the next multiple of 64 past the logical end
            n = self.value.numel()
            buf = torch.zeros((n + 63) // 64 * 64, device=self.value.device, dtype=self.value.dtype)
            self.grad = buf[:n].view(self.value.shape)
        return self.grad


class LARSOptimizer:
    """Layer-wise Adaptive Rate Scaling (tf2/lars_optimizer.py:25)."""

    def __init__(self,
                 learning_rate,
                 momentum=0.9,
                 use_nesterov=False,
                 weight_decay=0.0,
                 exclude_from_weight_decay=None,
                 exclude_from_layer_adaptation=None,
                 classic_momentum=True,
                 eeta=EETA_DEFAULT,
                 name="LARSOptimizer"):
        self.name = name
        self.learning_rate = learning_rate          # float or callable(step) (a schedule)
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.use_nesterov = use_nesterov
        self.classic_momentum = classic_momentum
        self.eeta = eeta
        self.exclude_from_weight_decay = exclude_from_weight_decay
        # tf2/lars_optimizer.py:72-77
        if exclude_from_layer_adaptation:
            self.exclude_from_layer_adaptation = exclude_from_layer_adaptation
        else:
            self.exclude_from_layer_adaptation = exclude_from_weight_decay
        self.iterations = 0
        self._slots = {}
        self._key = None
        self._table = self._chunks = self._norms = None
        self._lr_dev = None

    # -- tf2/lars_optimizer.py:79-81
    def _create_slots(self, var_list):
        for v in var_list:
            if id(v) not in self._slots:
                self._slots[id(v)] = torch.zeros_like(v.value)

    def get_slot(self, var, slot_name='Momentum'):
        assert slot_name == 'Momentum'
        return self._slots[id(var)]

    # -- tf2/lars_optimizer.py:139-148
    def _use_weight_decay(self, param_name):
        if not self.weight_decay:
            return False
        if self.exclude_from_weight_decay:
            for r in self.exclude_from_weight_decay:
                if re.search(r, param_name) is not None:
                    return False
        return True

    # -- tf2/lars_optimizer.py:150-157
    def _do_layer_adaptation(self, param_name):
        if self.exclude_from_layer_adaptation:
            for r in self.exclude_from_layer_adaptation:
                if re.search(r, param_name) is not None:
                    return False
        return True

    def _build(self, variables, grads=None):
        self._create_slots(variables)
        grads = [v.grad for v in variables] if grads is None else grads
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
            table[4 * T + t] = (1 if self._use_weight_decay(v.name) else 0) | \
                               (2 if self._do_layer_adaptation(v.name) else 0)
            for off in range(0, v.value.numel(), chunk):
                chunks.append((t, off))
        dev = variables[0].value.device
        self._table = table.to(dev)
        self._chunks = torch.tensor(chunks, dtype=torch.int64).view(-1).to(dev)
        self._norms = torch.zeros(2 * len(chunks), dtype=torch.float64, device=dev)   # per-chunk partial norms
        self._num = (T, len(chunks))
        self._key = tuple((v.value.data_ptr(), g.data_ptr()) for v, g in zip(variables, grads))

    def current_lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def apply_gradients(self, grads_and_vars, lr_device=None):
        """tf2/run.py:622.  grads_and_vars: iterable of (grad tensor, Variable)."""
        pairs = [(g, v) for g, v in grads_and_vars if g is not None and v is not None]   # :84-85
        grads = [g for g, _ in pairs]
        variables = [v for _, v in pairs]
        key = tuple((v.value.data_ptr(), g.data_ptr()) for v, g in zip(variables, grads))
        if key != self._key:
            self._build(variables, grads)
        T, nchunks = self._num
        lr = self.current_lr()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib().lars_multi_tensor(
            ctypes.c_void_p(self._table.data_ptr()), T, ctypes.c_void_p(self._chunks.data_ptr()), nchunks,
            ctypes.c_void_p(lr_device.data_ptr()) if lr_device is not None else None, lr,
            float(self.momentum), float(self.weight_decay or 0.0), float(self.eeta),
            int(bool(self.classic_momentum)), int(bool(self.use_nesterov)),
            ctypes.c_void_p(self._norms.data_ptr()), stream)
        self.iterations += 1

    def get_config(self):
        return {
            "name": self.name,
            "learning_rate": self.learning_rate if not callable(self.learning_rate) else 'schedule',
            "momentum": self.momentum,
            "classic_momentum": self.classic_momentum,
            "weight_decay": self.weight_decay,
            "eeta": self.eeta,
            "use_nesterov": self.use_nesterov,
        }
