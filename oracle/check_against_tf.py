"""Pins the oracle against the reference ITSELF when TensorFlow is importable.  TEST INFRASTRUCTURE ONLY.

TensorFlow is not installed in the build image (no network), so in this repository's CI every check
below is skipped; what runs instead is the reference's own source on a numpy stand-in for TensorFlow
(oracle/tfshim.py, tests/golden/make_reference_golden.py, tests/test_reference_pin.py -- see oracle/__init__.py).  On a machine that has
`tensorflow` (a Keras-2-era release: needs tf.keras.optimizers.legacy and
tf.keras.layers.experimental.SyncBatchNormalization) and `absl-py`, and the reference checkout at
REFERENCE (default /root/reference), run

    python -m oracle.check_against_tf            # prints one line per check, exit code 1 on mismatch

or `pytest tests/test_oracle.py -k tensorflow`.  Each check executes the reference function named in
its docstring and compares the oracle's float64 restatement with it on seeded random inputs
(tolerance 1e-5 relative: the reference computes in float32).
"""
import importlib
import os
import sys

import numpy as np

REFERENCE = os.environ.get('SIMCLR_REFERENCE', '/root/reference')


def available():
    """True when the reference can be executed here."""
    if not os.path.isdir(os.path.join(REFERENCE, 'tf2')):
        return False
    try:
        importlib.import_module('tensorflow')
        importlib.import_module('absl.flags')
    except Exception:       # noqa: any import failure means "cannot run the reference"
        return False
    return True


_loaded = {}


def _reference():
    """Imports tf2/{objective,lars_optimizer,resnet}.py from the reference checkout.  Those modules read
    absl FLAGS that tf2/run.py defines; run.py itself drags in tensorflow_datasets, so the handful of
    flags the three modules touch are defined here with run.py's defaults (tf2/run.py:49-53,106-126,208-226)."""
    if _loaded:
        return _loaded
    from absl import flags
    F = flags.FLAGS
    defs = [('global_bn', True, flags.DEFINE_boolean), ('batch_norm_decay', 0.9, flags.DEFINE_float),
            ('sk_ratio', 0.0, flags.DEFINE_float), ('se_ratio', 0.0, flags.DEFINE_float),
            ('train_mode', 'pretrain', flags.DEFINE_string), ('fine_tune_after_block', -1, flags.DEFINE_integer)]
    for name, default, fn in defs:
        if name not in F:
            fn(name, default, 'see tf2/run.py')
    if not F.is_parsed():
        F(['check_against_tf'])
    sys.path.insert(0, os.path.join(REFERENCE, 'tf2'))
    try:
        for m in ('objective', 'lars_optimizer', 'resnet'):
            _loaded[m] = importlib.import_module(m)
    finally:
        sys.path.pop(0)
    _loaded['flags'] = F
    return _loaded


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def check_contrastive_loss(n=24, d=32, temperature=0.1, hidden_norm=True, seed=0):
    """tf2/objective.py:35-89 (single replica) incl. its gradient (tape.gradient, tf2/run.py:621)."""
    import tensorflow as tf
    from oracle import ntxent as ont
    ref = _reference()['objective']
    h = np.random.default_rng(seed).standard_normal((2 * n, d)).astype(np.float32)
    x = tf.constant(h)
    with tf.GradientTape() as tape:
        tape.watch(x)
        loss, logits_ab, labels = ref.add_contrastive_loss(x, hidden_norm=hidden_norm, temperature=temperature)
    g = tape.gradient(loss, x).numpy()
    o_loss, o_logits, o_labels = ont.add_contrastive_loss(h, hidden_norm, temperature)
    _, o_grads = ont.contrastive_loss_and_grad([h], hidden_norm, temperature)
    return dict(loss=_rel(o_loss, loss.numpy()), logits_ab=_rel(o_logits, logits_ab.numpy()),
                labels=float(np.abs(o_labels - labels.numpy()).max()), grad=_rel(o_grads[0], g))


def check_lars(seed=0):
    """tf2/lars_optimizer.py:83-157: every momentum variant, the name filters and the zero-norm branches."""
    import tensorflow as tf
    from oracle import lars as olars
    ref = _reference()['lars_optimizer']
    rng = np.random.default_rng(seed)
    excl = ['batch_normalization', 'bias', 'head_supervised']
    out = {}
    for classic in (True, False):
        for nest in (False, True):
            named = [('conv2d/kernel', (3, 3, 4, 8)), ('batch_normalization/gamma', (8,)),
                     ('head_supervised/linear_layer/dense/bias', (5,)), ('zero/kernel', (6,))]
            vs, gs, ws = [], [], []
            for name, shp in named:
                w = (rng.standard_normal(shp) * 0.05).astype(np.float32)
                if name.startswith('zero'):
                    w[:] = 0
                ws.append(w)
                vs.append(tf.Variable(w, name=name))
                gs.append((rng.standard_normal(shp) * 1e-3).astype(np.float32))
            opt = ref.LARSOptimizer(0.3, momentum=0.9, use_nesterov=nest, weight_decay=1e-4, classic_momentum=classic,
                                    exclude_from_weight_decay=excl)
            for step in range(2):          # the second step exercises a non-zero Momentum slot
                before = [v.numpy().copy() for v in vs]
                slots = [opt.get_slot(v, 'Momentum').numpy().copy() if step else np.zeros_like(b) for v, b in zip(vs, before)]
                opt.apply_gradients(zip([tf.constant(g) for g in gs], vs))
                for v, g, b, m in zip(vs, gs, before, slots):
                    nw, nv = olars.lars_apply(v.name, b, g, m, 0.3, momentum=0.9, use_nesterov=nest, weight_decay=1e-4,
                                              classic_momentum=classic, exclude_from_weight_decay=excl)
                    key = '%s classic=%d nesterov=%d step%d' % (v.name, classic, nest, step)
                    out[key] = max(_rel(nw, v.numpy()) if np.abs(nw).max() > 0 else float(np.abs(v.numpy()).max()),
                                   _rel(nv, opt.get_slot(v, 'Momentum').numpy()))
    return out


def check_batch_norm_relu(seed=0):
    """tf2/resnet.py:31-78 with global_bn=False (Keras non-fused BatchNormalization; one replica): training-mode output,
    the moving-average update (biased variance) and the inference-mode output."""
    import tensorflow as tf
    import torch
    from oracle.model_torch import Builder, Config
    refs = _reference()
    refs['flags'].global_bn = False
    try:
        x = np.random.default_rng(seed).standard_normal((6, 5, 4, 8)).astype(np.float32) * 2 + 0.5
        layer = refs['resnet'].BatchNormRelu(relu=True)
        y_tf = layer(tf.constant(x), training=True).numpy()
        mm = layer.bn.moving_mean.numpy()
        mv = layer.bn.moving_variance.numpy()
        y_eval = layer(tf.constant(x), training=False).numpy()
        b = Builder(Config(global_bn=False), dtype=torch.float64)
        y = b.batch_norm_relu(torch.from_numpy(x).double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
        st = {k.rsplit('/', 1)[1]: v.numpy() for k, v in b.new_state.items()}
        b2 = Builder(Config(global_bn=False), params=b.params, dtype=torch.float64,
                     state={k: v.double() for k, v in b.new_state.items()})
        b2.training = False
        y2 = b2.batch_norm_relu(torch.from_numpy(x).double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
        return dict(train_out=_rel(y, y_tf), moving_mean=_rel(st['moving_mean:0'], mm),
                    moving_variance=_rel(st['moving_variance:0'], mv), eval_out=_rel(y2, y_eval))
    finally:
        refs['flags'].global_bn = True


def check_conv2d_fixed_padding(seed=0):
    """tf2/resnet.py:160-208: explicit (k-1)//2 padding + VALID at stride 2, SAME at stride 1, odd and even sizes."""
    import tensorflow as tf
    import torch
    from oracle.model_torch import Builder, Config
    ref = _reference()['resnet']
    rng = np.random.default_rng(seed)
    out = {}
    for (h, k, s) in [(9, 3, 2), (8, 3, 2), (7, 1, 2), (6, 3, 1), (10, 7, 2)]:
        x = rng.standard_normal((2, h, h, 4)).astype(np.float32)
        layer = ref.Conv2dFixedPadding(filters=6, kernel_size=k, strides=s)
        y_tf = layer(tf.constant(x), training=True).numpy()
        w = layer.conv2d.kernel.numpy()                    # HWIO
        b = Builder(Config(), dtype=torch.float64)
        b.init = False
        b.params = {'conv2d_fixed_padding/conv2d/kernel:0': torch.from_numpy(w).double()}
        y = b.conv2d_fixed_padding(torch.from_numpy(x).double().permute(0, 3, 1, 2), 6, k, s).permute(0, 2, 3, 1).numpy()
        out['h%d k%d s%d' % (h, k, s)] = _rel(y, y_tf) if y.shape == y_tf.shape else 1.0
    return out


def run_all(tol=1e-5):
    """Returns (ok, {check: {item: relative error}})."""
    res = dict(contrastive_loss=check_contrastive_loss(), contrastive_loss_T1_nonorm=check_contrastive_loss(hidden_norm=False, temperature=1.0),
               lars=check_lars(), batch_norm_relu=check_batch_norm_relu(), conv2d_fixed_padding=check_conv2d_fixed_padding())
    ok = all(v <= tol for d in res.values() for v in d.values())
    return ok, res


if __name__ == '__main__':
    if not available():
        print('tensorflow / absl / %s not available: the reference cannot be executed here (see tests/test_reference_pin.py for the stand-in run)' % REFERENCE)
        sys.exit(0)
    ok, res = run_all()
    for name, d in res.items():
        for k, v in d.items():
            print('%-28s %-60s %.3e' % (name, k, v))
    sys.exit(0 if ok else 1)
