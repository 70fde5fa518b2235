"""numpy stand-in for the slice of the TensorFlow / Keras / absl API that the reference's tf2/*.py touch.  TEST INFRASTRUCTURE ONLY.

Why: TensorFlow cannot be installed in the build image, so the reference (pure Python on top of TensorFlow) could never be
executed next to the oracle and the oracle stayed "checked against itself".  With this module installed as `tensorflow` /
`tensorflow.compat.v2` / `absl` the reference's OWN SOURCE FILES (/root/reference/tf2/{objective,lars_optimizer,model,metrics,
resnet,data_util}.py) import and run unmodified: their control flow, masks, label layout, concatenation order, name filters,
momentum variants, schedule arithmetic, block wiring and variable naming execute as written; only the primitives underneath
are numpy (float64).  That pins the oracle's restatement of the reference's LOGIC.  It does not pin TensorFlow's kernels: every
primitive below is a few lines of textbook numpy following the documented TensorFlow semantics named in its docstring
(tf.nn.conv2d SAME/VALID padding, Keras BatchNormalization training mode with biased variance, tf.keras.experimental.CosineDecay,
avg-pool SAME counting valid elements only, ...).  DESIGN.md section 5 states exactly this split.

Conventions: every float dtype is float64 (a `tf.cast(x, tf.float32)` stays float64: the pin is about structure, not rounding),
every integer dtype int64.  Tensors are `T`, an ndarray subclass with a TensorFlow-like `.shape` (`.ndims`, `.as_list()`),
`.numpy()` and NON-mutating augmented assignment (TensorFlow tensors are immutable: `grad += wd * param` rebinds).

Used by tests/golden/make_reference_golden.py (writes the fixtures) and tests/test_reference_pin.py (replays them when
/root/reference is present).  Nothing under simclr_amd/ imports it.
"""
import contextlib
import math
import re
import sys
import threading
import types

import numpy as np


# ------------------------------------------------------------------------------------------------ tensors
class TShape(tuple):
    @property
    def ndims(self):
        return len(self)

    rank = ndims

    def as_list(self):
        return list(self)


class T(np.ndarray):
    __array_priority__ = 1000.0

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    @property
    def shape(self):
        return TShape(np.ndarray.shape.__get__(self, type(self)))

    @shape.setter
    def shape(self, v):
        np.ndarray.shape.__set__(self, v)

    def get_shape(self):
        return self.shape

    def numpy(self):
        return np.array(self, copy=True).view(np.ndarray)

    # immutable-tensor semantics for `x += y` and friends
    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o


def _a(x):
    """plain ndarray view of anything tensor-like (T, Variable, list, scalar)"""
    if isinstance(x, Variable):
        return np.asarray(x.value).view(np.ndarray)
    return np.asarray(x).view(np.ndarray)


def _f(x):
    a = _a(x)
    return a.astype(np.float64) if a.dtype.kind == 'f' and a.dtype != np.float64 else a


def _ints(x):
    """shape-like argument (list that may hold T / numpy ints, or an int tensor) -> list of python ints"""
    if isinstance(x, (list, tuple)):
        out = []
        for v in x:
            if isinstance(v, (list, tuple, np.ndarray)) and np.ndim(v) > 0:
                out.extend(int(u) for u in np.ravel(_a(v)))
            else:
                out.append(int(v))
        return out
    return [int(u) for u in np.ravel(_a(x))]


def _axis(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple, np.ndarray)):
        return tuple(int(a) for a in np.ravel(_a(axis)))
    return int(axis)


class DType:
    def __init__(self, name, np_dtype):
        self.name = name
        self.np = np_dtype
        self.base_dtype = self

    def __repr__(self):
        return 'tf.' + self.name


float32 = DType('float32', np.float64)      # served in float64 (module docstring)
float64 = DType('float64', np.float64)
bfloat16 = DType('bfloat16', np.float64)
int32 = DType('int32', np.int64)
int64 = DType('int64', np.int64)
uint32 = DType('uint32', np.int64)
bool_ = DType('bool', np.bool_)


class Variable:
    """tf.Variable as the reference uses it: .name, .device, .dtype.base_dtype, .assign(), arithmetic, .numpy()."""

    def __init__(self, value, name='Variable', trainable=True, dtype=None):
        self.value = np.array(_f(value), copy=True)
        self.name = name if name.endswith(':0') else name + ':0'
        self.trainable = trainable
        self.device = ''
        self.dtype = float32
        CREATED_VARIABLES.append(self)

    @property
    def shape(self):
        return TShape(self.value.shape)

    def assign(self, v, use_locking=False):
        self.value = np.array(_f(v), copy=True).reshape(self.value.shape)
        return self

    def numpy(self):
        return np.array(self.value, copy=True)

    def __array__(self, dtype=None, copy=None):
        return self.value if dtype is None else self.value.astype(dtype)

    def __add__(self, o): return T(self.value + _a(o))
    def __radd__(self, o): return T(_a(o) + self.value)
    def __sub__(self, o): return T(self.value - _a(o))
    def __rsub__(self, o): return T(_a(o) - self.value)
    def __mul__(self, o): return T(self.value * _a(o))
    def __rmul__(self, o): return T(_a(o) * self.value)
    def __truediv__(self, o): return T(self.value / _a(o))
    def __neg__(self): return T(-self.value)


CREATED_VARIABLES = []          # every Variable in creation order (weight injection / name comparison by the harness)


# ------------------------------------------------------------------------------------------------ tf.* functions
def constant(x, dtype=None, shape=None):
    t = cast(x, dtype) if dtype is not None else T(_f(x))
    return T(np.reshape(_a(t), _ints(shape))) if shape is not None else t


convert_to_tensor = constant


def cast(x, dtype):
    return T(_a(x).astype(dtype.np))


def identity(x, name=None):
    return T(_a(x))


def stop_gradient(x):
    return T(_a(x))


def shape(x):
    return T(np.array(_a(x).shape, dtype=np.int64))


def size(x):
    return int(_a(x).size)


def range(*args, **kw):      # noqa: A001 (mirrors tf.range)
    return T(np.arange(*[int(v) for v in args]))


def repeat(x, repeats):
    return T(np.repeat(_a(x), int(repeats)))


def one_hot(indices, depth):
    """tf.one_hot: float rows with a single 1 (all zeros for an index outside [0, depth))"""
    idx = _a(indices).astype(np.int64)
    out = np.zeros(idx.shape + (int(depth),), np.float64)
    ok = (idx >= 0) & (idx < int(depth))
    np.put_along_axis(out, np.where(ok, idx, 0)[..., None], ok[..., None].astype(np.float64), axis=-1)
    return T(out)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _f(a), _f(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return T(a @ b)


def concat(values, axis):
    return T(np.concatenate([_a(v) for v in values], axis=int(axis)))


def stack(values, axis=0):
    return T(np.stack([_a(v) for v in values], axis=int(axis)))


def split(value, num_or_size_splits, axis=0):
    """tf.split: an int = that many equal parts; a list / tensor = the sizes of the parts"""
    a = _a(value)
    if np.ndim(num_or_size_splits) == 0:
        return [T(p) for p in np.split(a, int(num_or_size_splits), axis=int(axis))]
    sizes = _ints(num_or_size_splits)
    assert sum(sizes) == a.shape[int(axis)], (sizes, a.shape)
    return [T(p) for p in np.split(a, np.cumsum(sizes)[:-1], axis=int(axis))]


def reshape(x, shp):
    return T(np.reshape(_a(x), _ints(shp)))


def expand_dims(x, axis):
    return T(np.expand_dims(_a(x), int(axis)))


def squeeze(x, axis=None):
    return T(np.squeeze(_a(x), axis=_axis(axis)))


def tile(x, multiples):
    return T(np.tile(_a(x), _ints(multiples)))


def pad(x, paddings):
    """tf.pad, CONSTANT mode, zeros"""
    return T(np.pad(_a(x), [tuple(int(v) for v in p) for p in paddings]))


def transpose(x, perm=None):
    return T(np.transpose(_a(x), perm))


def reduce_mean(x, axis=None, keepdims=False):
    return T(np.mean(_f(x), axis=_axis(axis), keepdims=keepdims))


def reduce_sum(x, axis=None, keepdims=False):
    return T(np.sum(_f(x), axis=_axis(axis), keepdims=keepdims))


def reduce_max(x, axis=None, keepdims=False):
    return T(np.max(_a(x), axis=_axis(axis), keepdims=keepdims))


def reduce_min(x, axis=None, keepdims=False):
    return T(np.min(_a(x), axis=_axis(axis), keepdims=keepdims))


def add_n(xs):
    out = _f(xs[0])
    for v in xs[1:]:
        out = out + _f(v)
    return T(out)


def multiply(a, b):
    return T(_f(a) * _f(b))


def equal(a, b):
    return T(_a(a) == _a(b))


def greater(a, b):
    return T(_a(a) > _a(b))


def less(a, b):
    return T(_a(a) < _a(b))


def logical_and(a, b):
    return T(np.logical_and(_a(a), _a(b)))


def where(cond, x, y):
    return T(np.where(_a(cond), _f(x), _f(y)))


def argmax(x, axis=None, **kw):
    """tf.argmax: axis defaults to 0; ties -> the smallest index (numpy agrees)"""
    return T(np.argmax(_a(x), axis=0 if axis is None else int(axis)))


def norm(x, ord=2, axis=None):      # noqa: A002
    """tf.norm with axis=None treats the tensor as one vector; ord=2 -> sqrt(sum x^2)"""
    assert ord in (2, 'euclidean') and axis is None
    a = _f(x)
    return T(np.sqrt(np.sum(a * a)))


def exp(x):
    return T(np.exp(_f(x)))


def pow(x, y):      # noqa: A001
    return T(np.power(_f(x), _f(y)))


def sqrt(x):
    return T(np.sqrt(_f(x)))


def sigmoid(x):
    return T(1.0 / (1.0 + np.exp(-_f(x))))


def clip_by_value(x, lo, hi):
    return T(np.clip(_f(x), lo, hi))


def zeros_like(x):
    return T(np.zeros_like(_f(x)))


def scatter_nd(indices, updates, shape):      # noqa: A002
    """tf.scatter_nd for the one form the reference uses (objective.py:113-116): indices [[i]], updates [tensor]"""
    out = np.zeros(_ints(shape), np.float64)
    for idx, upd in zip(indices, updates):
        out[tuple(int(v) for v in idx)] += _f(upd)
    return T(out)


def no_op():
    return None


def group(*ops):
    return None


@contextlib.contextmanager
def name_scope(name):
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


_SCOPE = []


class _Initializer:
    def __init__(self, kind, **kw):
        self.kind, self.kw = kind, kw

    def __call__(self, shp, dtype=None):
        # values are injected by the harness; only zeros / ones carry meaning (BatchNormRelu's init_zero)
        return np.ones(shp) if self.kind == 'ones' else np.zeros(shp)


def zeros_initializer():
    return _Initializer('zeros')


def ones_initializer():
    return _Initializer('ones')


# ------------------------------------------------------------------------------------------------ tf.math / tf.nn
def _l2_normalize(x, axis=None, epsilon=1e-12):
    """tf.math.l2_normalize: x * rsqrt(max(sum(x^2, axis), epsilon))"""
    a = _f(x)
    return T(a / np.sqrt(np.maximum(np.sum(a * a, axis=_axis(axis), keepdims=True), epsilon)))


def _log_softmax(z, axis=-1):
    z = z - np.max(z, axis=axis, keepdims=True)
    return z - np.log(np.sum(np.exp(z), axis=axis, keepdims=True))


def _softmax(x, axis=-1):
    return T(np.exp(_log_softmax(_f(x), int(axis))))


def _softmax_xent(labels, logits, axis=-1):
    """tf.nn.softmax_cross_entropy_with_logits: -sum(labels * log_softmax(logits))"""
    return T(-np.sum(_f(labels) * _log_softmax(_f(logits), int(axis)), axis=int(axis)))


def _same_pad(n, k, s):
    """TensorFlow SAME: out = ceil(n / s), total = max((out - 1) s + k - n, 0), before = total // 2"""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _windows(x, kh, kw, sh, sw, padding, fill=0.0):
    """x [N,H,W,C] -> windows [N,H',W',C,kh,kw] and a validity mask [1,H',W',1,kh,kw] (False on SAME padding)"""
    n, h, w, c = x.shape
    if str(padding).upper() == 'SAME':
        (pt, pb), (pl, pr) = _same_pad(h, kh, sh), _same_pad(w, kw, sw)
    else:
        pt = pb = pl = pr = 0
    xp = np.pad(x, [(0, 0), (pt, pb), (pl, pr), (0, 0)], constant_values=fill)
    mp = np.pad(np.ones((1, h, w, 1), bool), [(0, 0), (pt, pb), (pl, pr), (0, 0)], constant_values=False)
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))[:, ::sh, ::sw]
    msk = np.lib.stride_tricks.sliding_window_view(mp, (kh, kw), axis=(1, 2))[:, ::sh, ::sw]
    return win, msk


def _conv2d(x, kernel, strides, padding):
    """tf.nn.conv2d, NHWC input, HWIO kernel (cross-correlation, as TensorFlow defines it)"""
    x, k = _f(x), _f(kernel)
    kh, kw = k.shape[:2]
    sh, sw = (strides, strides) if np.ndim(strides) == 0 else (strides[-3], strides[-2]) if len(strides) == 4 else tuple(strides)
    win, _ = _windows(x, kh, kw, int(sh), int(sw), padding)
    return T(np.einsum('nhwcij,ijcf->nhwf', win, k, optimize=True))


def _depthwise_conv2d(x, filt, strides, padding):
    """tf.nn.depthwise_conv2d with channel multiplier 1: filter [kh, kw, C, 1]"""
    x, k = _f(x), _f(filt)
    win, _ = _windows(x, k.shape[0], k.shape[1], int(strides[1]), int(strides[2]), padding)
    return T(np.einsum('nhwcij,ijc->nhwc', win, k[..., 0], optimize=True))


def _max_pool(x, ksize, strides, padding):
    k = ksize if np.ndim(ksize) == 0 else ksize[1] if len(ksize) == 4 else ksize[0]
    s = strides if np.ndim(strides) == 0 else strides[1] if len(strides) == 4 else strides[0]
    win, _ = _windows(_f(x), int(k), int(k), int(s), int(s), padding, fill=-np.inf)
    return T(win.max(axis=(-1, -2)))


def _avg_pool(x, k, s, padding):
    """tf.nn.avg_pool: SAME padding divides by the number of VALID elements of each window"""
    win, msk = _windows(_f(x), int(k), int(k), int(s), int(s), padding)
    return T(win.sum(axis=(-1, -2)) / msk.sum(axis=(-1, -2)))


# ------------------------------------------------------------------------------------------------ tf.random / tf.image (scripted)
# data_util.py draws its randomness from TensorFlow.  Here every draw is SCRIPTED: the harness puts the outcomes into SCRIPT before it
# calls preprocess_for_train, each stand-in hands out the scripted value and RECORDS the bounds / arguments the reference asked with
# (SCRIPT['asked']), so that both the arithmetic given the draws and the ranges the reference draws from can be compared.  The pixel
# kernels (bicubic resize, contrast / saturation / hue, grayscale) are oracle/augment.py's restatements of TensorFlow kernels that are
# not under /root/reference: this part of the pin covers the reference's COMPOSITION (order, gates, clipping, strength scaling,
# crop-shape arithmetic), not those kernels.
SCRIPT = {}


def _asked(kind, *args):
    SCRIPT.setdefault('asked', []).append((kind,) + tuple(float(a) for a in args))


def _random_uniform(shape, minval=0, maxval=1, dtype=None, **kw):
    assert list(shape) == [], 'scripted draws are scalars'
    lo, hi = float(_a(minval)), float(_a(maxval))
    if (lo, hi) == (0.0, 1.0):
        return T(np.float64(SCRIPT['unit'].pop(0)))            # the gates of random_apply, in call order
    _asked('uniform', lo, hi)
    return T(np.float64(SCRIPT['uniform'].pop(0)))              # brightness factor / blur sigma, in call order


def _random_shuffle(x):
    perm = [int(v) for v in SCRIPT['perm']]
    assert sorted(perm) == sorted(int(v) for v in _a(x))
    return T(np.array(perm))


def cond(pred, true_fn, false_fn):
    return true_fn() if bool(_a(pred)) else false_fn()


def maximum(a, b):
    return T(np.maximum(_f(a), _f(b)))


def unstack(x, axis=0):
    a = _a(x)
    return [a[i] for i in np.arange(a.shape[int(axis)])] if int(axis) == 0 else [T(v) for v in np.moveaxis(a, int(axis), 0)]


def _image_ns():
    from oracle import augment as oa

    def sample_distorted_bounding_box(image_size, bounding_boxes, min_object_covered=0.1, aspect_ratio_range=(0.75, 1.33),
                                      area_range=(0.05, 1.0), max_attempts=100, use_image_if_no_bounding_boxes=False, **kw):
        _asked('bbox', min_object_covered, aspect_ratio_range[0], aspect_ratio_range[1], area_range[0], area_range[1], max_attempts,
               float(use_image_if_no_bounding_boxes), *np.ravel(_a(bounding_boxes)))
        y, x, h, w = (int(v) for v in SCRIPT['crop'])
        return T(np.array([y, x, 0])), T(np.array([h, w, -1])), None

    def crop_to_bounding_box(image, oy, ox, th, tw):
        _asked('cropbox', int(oy), int(ox), int(th), int(tw))
        return T(_a(image)[int(oy):int(oy) + int(th), int(ox):int(ox) + int(tw)])

    def resize(images, size, method=None):
        assert method == 'bicubic'
        return [T(oa.resize_bicubic(_f(im), int(size[0]), int(size[1]))) for im in images]

    def random_flip_left_right(image):
        return T(_a(image)[:, ::-1]) if SCRIPT['flip'] else T(_a(image))

    def _rand_op(kind, fn):
        def op(image, lower, upper=None):
            lo, hi = (-float(lower), float(lower)) if upper is None else (float(lower), float(upper))      # random_hue(max_delta)
            _asked(kind, lo, hi)
            return T(fn(_f(image), float(SCRIPT[kind])))
        return op

    def rgb_to_grayscale(image):
        return T(oa.to_grayscale(_f(image))[..., :1])

    def convert_image_dtype(image, dtype=None):
        a = _a(image)
        return T(a.astype(np.float64) * (1.0 / 255.0)) if a.dtype == np.uint8 else T(a.astype(np.float64))

    return _ns('tensorflow.image', sample_distorted_bounding_box=sample_distorted_bounding_box, crop_to_bounding_box=crop_to_bounding_box,
               resize=resize, ResizeMethod=types.SimpleNamespace(BICUBIC='bicubic'), random_flip_left_right=random_flip_left_right,
               random_contrast=_rand_op('contrast', oa.adjust_contrast), random_saturation=_rand_op('saturation', oa.adjust_saturation),
               random_hue=lambda image, max_delta: _rand_op('hue', oa.adjust_hue)(image, max_delta),
               rgb_to_grayscale=rgb_to_grayscale, convert_image_dtype=convert_image_dtype)


# ------------------------------------------------------------------------------------------------ tf.distribute (replica emulation)
_TLS = threading.local()


class _ReplicaContext:
    def __init__(self, rid, bus):
        self.replica_id_in_sync_group = rid
        self._bus = bus

    def all_reduce(self, op, tensor):
        assert op == 'SUM'
        b = self._bus
        b['slots'][self.replica_id_in_sync_group] = _f(tensor)
        b['barrier'].wait()
        out = b['slots'][0].copy()
        for s in b['slots'][1:]:
            out = out + s
        b['barrier'].wait()
        return T(out)


class Strategy:
    """Stand-in for tf.distribute.Strategy: `run(fn, per_replica_args)` executes fn once per replica on threads that meet
    in all_reduce -- the lockstep a TPU/GPU strategy provides."""

    def __init__(self, num_replicas):
        self.num_replicas_in_sync = num_replicas

    def run(self, fn, per_replica_args):
        R = self.num_replicas_in_sync
        bus = dict(slots=[None] * R, barrier=threading.Barrier(R))
        out, err = [None] * R, []

        def work(i):
            _TLS.ctx = _ReplicaContext(i, bus)
            try:
                out[i] = fn(*per_replica_args[i])
            except BaseException as e:      # noqa: surfaced below
                err.append(e)
                bus['barrier'].abort()
            finally:
                _TLS.ctx = None
        ths = [threading.Thread(target=work, args=(i,)) for i in np.arange(R)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            raise err[0]
        return out


def _get_replica_context():
    return getattr(_TLS, 'ctx', None) or _ReplicaContext(0, dict(slots=[None], barrier=threading.Barrier(1)))


# ------------------------------------------------------------------------------------------------ tf.keras
_UIDS = {}


def reset_uids():
    """new Keras 'graph': layer auto-names start again at conv2d, conv2d_1, ..."""
    _UIDS.clear()
    del CREATED_VARIABLES[:]


def _snake(name):
    """keras.utils.generic_utils.to_snake_case"""
    inter = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    s = re.sub('([a-z])([A-Z])', r'\1_\2', inter).lower()
    return 'private' + s if s[0] == '_' else s


def _unique(base, zero_based=False):
    """keras.backend.unique_object_name: conv2d, conv2d_1, ... (zero_based -- Models -- : model, model_1 the same way)"""
    n = _UIDS.get(base, 0)
    _UIDS[base] = n + 1
    return base if n == 0 else '%s_%d' % (base, n)


class Layer:
    _zero_based = False

    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        assert not kwargs, 'unsupported Layer kwargs: %r' % (kwargs,)
        self._name = name if name else _unique(_snake(type(self).__name__), self._zero_based)
        self.trainable = trainable
        self.built = False
        self._weights = []

    @property
    def name(self):
        return self._name

    def build(self, input_shape):
        self.built = True

    def add_weight(self, name, shape, initializer=None, trainable=True):      # noqa: A002
        init = initializer if initializer is not None else _Initializer('zeros')
        v = Variable(init(tuple(int(s) for s in shape)), name='/'.join(_SCOPE + [name]), trainable=trainable and self.trainable)
        self._weights.append(v)
        return v

    def __call__(self, *args, **kwargs):
        with name_scope(self.name):
            if not self.built:
                self.build(args[0].shape if hasattr(args[0], 'shape') else None)
                self.built = True
            return self.call(*args, **kwargs)

    def _sublayers(self, seen):
        for v in list(vars(self).values()):
            items = v if isinstance(v, (list, tuple)) else [v]
            for it in items:
                if isinstance(it, Layer) and id(it) not in seen:
                    seen.add(id(it))
                    yield it
                    yield from it._sublayers(seen)

    @property
    def variables(self):
        out = list(self._weights)
        for l in self._sublayers({id(self)}):
            out.extend(l._weights)
        return out

    weights = variables

    @property
    def trainable_variables(self):
        return [v for v in self.variables if v.trainable]

    trainable_weights = trainable_variables


class Model(Layer):
    _zero_based = True


def _pair(v):
    return (int(v), int(v)) if np.ndim(v) == 0 else (int(v[0]), int(v[1]))


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding='valid', use_bias=True, kernel_initializer=None,
                 data_format='channels_last', **kw):
        super().__init__(**kw)
        assert data_format == 'channels_last'
        self.filters, self.kernel_size, self.strides = filters, _pair(kernel_size), _pair(strides)
        self.padding, self.use_bias = padding, use_bias

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', self.kernel_size + (int(input_shape[-1]), int(self.filters)))
        self.bias = self.add_weight('bias', (int(self.filters),)) if self.use_bias else None
        self.built = True

    def call(self, inputs, training=None):
        y = _conv2d(inputs, self.kernel, self.strides, self.padding)
        return T(y + self.bias.value) if self.use_bias else y


class Dense(Layer):
    def __init__(self, units, kernel_initializer=None, use_bias=True, **kw):
        super().__init__(**kw)
        self.units, self.use_bias = units, use_bias

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', (int(input_shape[-1]), int(self.units)))
        self.bias = self.add_weight('bias', (int(self.units),)) if self.use_bias else None
        self.built = True

    def call(self, inputs, training=None):
        y = _f(inputs) @ self.kernel.value
        return T(y + self.bias.value) if self.use_bias else T(y)


class BatchNormalization(Layer):
    """Keras BatchNormalization, non-fused: training -> batch mean / BIASED variance over every axis but `axis`, moving
    statistics updated as moving * momentum + batch * (1 - momentum); inference -> the moving statistics."""

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, fused=None, gamma_initializer=None, **kw):
        super().__init__(**kw)
        assert axis in (-1, 3, 1)
        self.axis, self.momentum, self.epsilon, self.center, self.scale = axis, momentum, epsilon, center, scale
        self.gamma_initializer = gamma_initializer or ones_initializer()

    def build(self, input_shape):
        c = int(input_shape[self.axis])
        self.gamma = self.add_weight('gamma', (c,), self.gamma_initializer) if self.scale else None
        self.beta = self.add_weight('beta', (c,), zeros_initializer()) if self.center else None
        self.moving_mean = self.add_weight('moving_mean', (c,), zeros_initializer(), trainable=False)
        self.moving_variance = self.add_weight('moving_variance', (c,), ones_initializer(), trainable=False)
        self.built = True

    def _moments(self, x, axes):
        mean = x.mean(axis=axes)
        return mean, ((x - mean) ** 2).mean(axis=axes)

    def call(self, inputs, training=None):
        x = _f(inputs)
        assert self.axis in (-1, x.ndim - 1)
        axes = tuple(np.arange(x.ndim - 1))
        if training:
            mean, var = self._moments(x, axes)
            self.moving_mean.assign(self.moving_mean.value * self.momentum + mean * (1 - self.momentum))
            self.moving_variance.assign(self.moving_variance.value * self.momentum + var * (1 - self.momentum))
        else:
            mean, var = self.moving_mean.value, self.moving_variance.value
        y = (x - mean) / np.sqrt(var + self.epsilon)
        if self.scale:
            y = y * self.gamma.value
        if self.center:
            y = y + self.beta.value
        return T(y)


class SyncBatchNormalization(BatchNormalization):
    """tf.keras.layers.experimental.SyncBatchNormalization: the same moments over the GLOBAL batch (sums all-reduced)"""

    def _moments(self, x, axes):
        ctx = _get_replica_context()
        n = float(np.prod([x.shape[a] for a in axes]))
        tot = _a(ctx.all_reduce('SUM', np.concatenate([[n], x.sum(axis=axes), (x * x).sum(axis=axes)])))
        c = x.shape[-1]
        mean = tot[1:1 + c] / tot[0]
        return mean, tot[1 + c:] / tot[0] - mean * mean


class MaxPooling2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding='valid', data_format='channels_last', **kw):
        super().__init__(**kw)
        self.pool_size, self.strides, self.padding = pool_size, strides or pool_size, padding

    def call(self, inputs, training=None):
        return _max_pool(inputs, self.pool_size, self.strides, self.padding)


class AveragePooling2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding='valid', data_format='channels_last', **kw):
        super().__init__(**kw)
        self.pool_size, self.strides, self.padding = pool_size, strides or pool_size, padding

    def call(self, inputs, training=None):
        return _avg_pool(inputs, self.pool_size, self.strides, self.padding)


class _LegacyOptimizer:
    """tf.keras.optimizers.legacy.Optimizer: hyper-parameters, slots, apply_gradients -> _resource_apply_dense per variable"""

    def __init__(self, name, **kw):
        self._name = name
        self._hyper = {}
        self._slots = {}
        self.iterations = 0

    def _set_hyper(self, k, v):
        self._hyper[k] = v

    def _get_hyper(self, k, dtype=None):
        v = self._hyper[k]
        return v(self.iterations) if callable(v) else v

    def _serialize_hyperparameter(self, k):
        return self._hyper[k]

    def add_slot(self, var, slot_name, initializer='zeros'):
        key = (id(var), slot_name)
        if key not in self._slots:
            self._slots[key] = Variable(np.zeros_like(var.value), name=var.name[:-2] + '/' + slot_name, trainable=False)
        return self._slots[key]

    def get_slot(self, var, slot_name):
        return self._slots[(id(var), slot_name)]

    def _fallback_apply_state(self, var_device, var_dtype):
        return {'lr_t': self._get_hyper('learning_rate')}

    def get_config(self):
        return {'name': self._name}

    def apply_gradients(self, grads_and_vars, name=None):
        gv = [(g, v) for g, v in grads_and_vars]
        self._create_slots([v for _, v in gv])
        for g, v in gv:
            self._resource_apply_dense(T(_f(g)) if g is not None else None, v)
        self.iterations += 1


class _CosineDecay:
    """tf.keras.experimental.CosineDecay(initial_learning_rate, decay_steps, alpha=0):
    step = min(step, decay_steps); lr = initial * ((1 - alpha) * 0.5 * (1 + cos(pi * step / decay_steps)) + alpha)"""

    def __init__(self, initial_learning_rate, decay_steps, alpha=0.0, name=None):
        self.lr0, self.decay_steps, self.alpha = initial_learning_rate, decay_steps, alpha

    def __call__(self, step):
        s = np.minimum(_f(np.asarray(step, np.float64)), float(self.decay_steps))
        cd = 0.5 * (1.0 + np.cos(math.pi * s / float(self.decay_steps)))
        return T(float(self.lr0) * ((1 - self.alpha) * cd + self.alpha))


class _CategoricalCrossentropy:
    def __init__(self, from_logits=False, reduction='auto'):
        assert from_logits and reduction == 'none'

    def __call__(self, labels, logits):
        return _softmax_xent(labels, logits)


class _Mean:
    """tf.keras.metrics.Mean as run.py uses it (one object shared by the emulated replicas, hence the lock)"""

    def __init__(self, name=None):
        self.name, self.total, self.count, self._lock = name, 0.0, 0, threading.Lock()

    def update_state(self, v):
        with self._lock:
            self.total += float(np.mean(_f(v)))
            self.count += 1

    def result(self):
        return T(np.float64(self.total / max(self.count, 1)))


class GradientTape:
    """tf.GradientTape for the one thing numpy can do with it: it cannot differentiate, so `gradient(loss, variables)` RECORDS the
    loss it was asked to differentiate (per emulated replica) and returns None gradients.  What tf2/run.py:557-622 composes into that
    loss, and which variables it hands to the optimizer, is thereby observable; the gradient values are the oracle's (torch autograd
    of the pinned forward)."""
    recorded = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def gradient(self, loss, variables):
        with _TAPE_LOCK:
            GradientTape.recorded[int(_get_replica_context().replica_id_in_sync_group)] = (float(_a(loss)), [v.name for v in variables])
        return [None for _ in variables]


_TAPE_LOCK = threading.Lock()


class RecordingOptimizer:
    """stands where tf2/run.py has its optimizer inside single_step: counts the call and keeps the variable order"""

    def __init__(self):
        self.iterations = 0
        self.applied = []

    def apply_gradients(self, grads_and_vars):
        with _TAPE_LOCK:
            self.applied.append([v.name for _, v in grads_and_vars])


# ------------------------------------------------------------------------------------------------ absl.flags
class _Flags:
    """absl.flags.FLAGS: a plain namespace the harness fills with tf2/run.py's defaults (run.py itself imports
    tensorflow_datasets and is not executed)."""

    def __contains__(self, k):
        return k in self.__dict__

    def is_parsed(self):
        return True


def _ns(name, **members):
    m = types.ModuleType(name)
    for k, v in members.items():
        setattr(m, k, v)
    return m


def install():
    """Registers the stand-in modules (tensorflow, tensorflow.compat.v2, absl.flags, absl.logging).  Returns (tf, FLAGS).
    Refuses to shadow a real TensorFlow."""
    if 'tensorflow' in sys.modules and not getattr(sys.modules['tensorflow'], '_SIMCLR_SHIM', False):
        raise RuntimeError('a real tensorflow is already imported: use oracle/check_against_tf.py instead of the stand-in')
    me = sys.modules[__name__]
    tf = _ns('tensorflow', _SIMCLR_SHIM=True)
    for k in dir(me):
        if not k.startswith('_') and k not in ('install', 'np', 'sys', 'math', 're', 'types', 'threading', 'contextlib'):
            setattr(tf, k, getattr(me, k))
    tf.bool = bool_
    tf.math = _ns('tensorflow.math', l2_normalize=_l2_normalize, log=lambda x: T(np.log(_f(x))), exp=exp, sqrt=sqrt)
    tf.nn = _ns('tensorflow.nn', softmax=_softmax, softmax_cross_entropy_with_logits=_softmax_xent,
                l2_loss=lambda v: T(np.sum(_f(v) ** 2) / 2.0), relu=lambda x: T(np.maximum(_f(x), 0.0)),
                conv2d=_conv2d, depthwise_conv2d=_depthwise_conv2d, max_pool=_max_pool)
    tf.distribute = _ns('tensorflow.distribute', get_replica_context=_get_replica_context, Strategy=Strategy,
                        ReduceOp=types.SimpleNamespace(SUM='SUM'))
    tf.random = _ns('tensorflow.random', uniform=_random_uniform, shuffle=_random_shuffle)
    tf.image = _image_ns()
    tf.math.rint = lambda x: T(np.rint(_f(x)))                 # round half to even, like tf.math.rint
    tf.summary = _ns('tensorflow.summary', scalar=lambda *a, **k: None, image=lambda *a, **k: None,
                     record_if=lambda cond_: contextlib.nullcontext())
    tf.logging = _ns('tensorflow.logging', info=lambda *a, **k: None)
    layers = _ns('tensorflow.keras.layers', Layer=Layer, Conv2D=Conv2D, Dense=Dense, BatchNormalization=BatchNormalization,
                 MaxPooling2D=MaxPooling2D, AveragePooling2D=AveragePooling2D,
                 experimental=types.SimpleNamespace(SyncBatchNormalization=SyncBatchNormalization))
    tf.keras = _ns('tensorflow.keras', layers=layers, models=types.SimpleNamespace(Model=Model),
                   initializers=types.SimpleNamespace(VarianceScaling=lambda **k: _Initializer('variance_scaling', **k),
                                                      RandomNormal=lambda **k: _Initializer('random_normal', **k)),
                   optimizers=types.SimpleNamespace(legacy=types.SimpleNamespace(Optimizer=_LegacyOptimizer),
                                                    schedules=types.SimpleNamespace(LearningRateSchedule=object),
                                                    SGD=None, Adam=None),
                   experimental=types.SimpleNamespace(CosineDecay=_CosineDecay),
                   losses=types.SimpleNamespace(CategoricalCrossentropy=_CategoricalCrossentropy,
                                                Reduction=types.SimpleNamespace(NONE='none')),
                   metrics=types.SimpleNamespace(Mean=_Mean))
    compat = _ns('tensorflow.compat', v2=tf, v1=tf)
    tf.compat = compat
    flags = _ns('absl.flags', FLAGS=_Flags())
    for k in ('DEFINE_float', 'DEFINE_integer', 'DEFINE_string', 'DEFINE_bool', 'DEFINE_boolean', 'DEFINE_enum'):
        setattr(flags, k, lambda *a, **kw: None)
    logging_ = _ns('absl.logging', info=lambda *a, **k: None, warning=lambda *a, **k: None)
    absl = _ns('absl', flags=flags, logging=logging_, app=_ns('absl.app'))
    sys.modules.update({'tensorflow': tf, 'tensorflow.compat': compat, 'tensorflow.compat.v2': tf, 'tensorflow.compat.v1': tf,
                        'absl': absl, 'absl.flags': flags, 'absl.logging': logging_, 'absl.app': absl.app})
    return tf, flags.FLAGS


def uninstall():
    for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.') or k == 'absl' or k.startswith('absl.')]:
        if getattr(sys.modules.get('tensorflow'), '_SIMCLR_SHIM', False) or k.startswith('absl'):
            sys.modules.pop(k, None)
