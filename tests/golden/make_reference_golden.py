"""Golden vectors produced by the REFERENCE'S OWN SOURCE FILES, executed in this container.  TEST INFRASTRUCTURE ONLY.

    python tests/golden/make_reference_golden.py            # rewrites tests/golden/reference_pin.npz (+ reference_pin.json)

TensorFlow is not installable here, so /root/reference/tf2/{objective,lars_optimizer,model,metrics,resnet,data_util}.py are
imported UNMODIFIED on top of oracle/tfshim.py -- a float64 numpy stand-in for the TensorFlow / Keras / absl calls they make
(each stand-in names the TensorFlow semantics it follows).  What these fixtures therefore pin: the reference's logic -- the
NT-Xent masks / label layout / concatenation order / cross-replica concat, the LARS branches and name filters, the schedule
and weight-decay arithmetic, the metrics, the blur filter, FixedPadding, and the wiring, variable names and initial values of
`resnet()` / `Model` (ResNet-18 CIFAR stem, ResNet-50, ResNet-50 + SK / ResNet-D) -- executed from the reference's code, not
restated.  What they do not pin: TensorFlow's own kernels.

`reference_cases(ref_dir)` runs the reference; `oracle_cases()` computes the same keys with oracle/*.py.  The test
(tests/test_reference_pin.py) compares oracle_cases() with the committed fixtures everywhere, and re-runs
reference_cases() against them wherever /root/reference exists.
"""
import importlib
import importlib.machinery
import importlib.util
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if HERE not in sys.path:
    sys.path.insert(0, HERE)          # recipe.py (pure numpy) sits next to this script

REFERENCE = os.environ.get('SIMCLR_REFERENCE', '/root/reference')
OUT_NPZ = os.path.join(HERE, 'reference_pin.npz')
OUT_JSON = os.path.join(HERE, 'reference_pin.json')

# tf2/run.py:37-238 defaults of the flags the imported modules read (run.py itself needs tensorflow_datasets)
FLAG_DEFAULTS = dict(global_bn=True, batch_norm_decay=0.9, sk_ratio=0.0, se_ratio=0.0, train_mode='pretrain', fine_tune_after_block=-1,
                     optimizer='lars', momentum=0.9, weight_decay=1e-6, warmup_epochs=10, train_batch_size=512,
                     learning_rate_scaling='linear', train_steps=0, train_epochs=100, proj_out_dim=128, proj_head_mode='nonlinear',
                     num_proj_layers=3, ft_proj_selector=0, resnet_depth=50, width_multiplier=1, image_size=224,
                     lineareval_while_pretraining=True, use_blur=True, hidden_norm=True, temperature=0.1, color_jitter_strength=1.0)

# ---- the case table (shared by both sides) ----------------------------------------------------------------------------
NTX = [dict(n=10, d=12, hidden_norm=True, temperature=0.1, seed=0), dict(n=7, d=9, hidden_norm=False, temperature=1.0, seed=1),
       dict(n=16, d=8, hidden_norm=True, temperature=0.5, seed=2)]
NTX_R = [dict(R=2, n=6, d=8, hidden_norm=True, temperature=0.1, seed=3), dict(R=3, n=4, d=6, hidden_norm=False, temperature=0.7, seed=4)]
LARS_NAMES = [('conv2d/kernel', (3, 3, 4, 8)), ('sync_batch_normalization/gamma', (8,)), ('head_supervised/linear_layer/dense/bias', (5,)),
              ('head_supervised/linear_layer/dense/kernel', (6, 5)), ('zero_weight/kernel', (6,)), ('zero_grad/kernel', (4, 3))]
LARS_EXCL = ['batch_normalization', 'bias', 'head_supervised']          # tf2/model.py:39-41
SCHED = [dict(scaling='linear', batch=512, warmup_epochs=10, train_epochs=100, train_steps=0, base_lr=0.3, num_examples=50000),
         dict(scaling='sqrt', batch=4096, warmup_epochs=10, train_epochs=800, train_steps=0, base_lr=0.075, num_examples=1281167),
         dict(scaling='linear', batch=256, warmup_epochs=0, train_epochs=5, train_steps=700, base_lr=1.0, num_examples=10000)]
SCHED_STEPS = [0, 1, 17, 400, 976, 977, 978, 5000, 9765, 9766, 20000, 250000, 300000]
CONV_CASES = [(9, 3, 2), (8, 3, 2), (7, 1, 2), (6, 3, 1), (10, 7, 2), (5, 1, 1)]
MODELS = [dict(tag='r18_cifar', depth=18, size=32, sk=0.0, batch=4, classes=10),
          dict(tag='r50', depth=50, size=48, sk=0.0, batch=3, classes=7),
          dict(tag='r50_sk', depth=50, size=48, sk=0.0625, batch=3, classes=7),
          # other flag values of tf2/run.py: width multiplier, depth 34, a two-layer head whose linear-eval input is the first
          # hidden layer (ft_proj_selector=1), a 64-wide projection, local BatchNorm, no linear-eval head
          dict(tag='r34_w2', depth=34, size=32, sk=0.0, batch=3, classes=5, width=2, num_proj_layers=2, ft_proj_selector=1, proj_out_dim=64),
          dict(tag='r18_localbn', depth=18, size=40, sk=0.0, batch=4, classes=6, global_bn=False, lineareval=False)]
# well-conditioned cases (tests/golden/recipe.py: variables by NAME from a pure-numpy recipe, image-like inputs, batch 16): fp32 arithmetic
# itself stays 5-10x inside north_star's 1e-5 / 1e-3 on them, so the device tests and bench.py gate them with FIXED tolerances
MODELS += [dict(tag=t, sk=0.0, recipe=True, **{k: v for k, v in c.items()}) for t, c in sorted(__import__('recipe').IMG_CASES.items())]
# the training step of tf2/run.py:557-622 (extracted from `main` by ast, see _single_step): (model tag, replicas)
STEPS = [('r18_cifar', 1), ('r18_cifar', 2), ('r50_sk', 1), ('r18_img', 1), ('r18_img', 2), ('r50_img', 1), ('r50', 1)]
# DIRECTIONAL derivatives <d loss / d variable, direction> of the reference's training step by CENTRAL DIFFERENCES of the reference's own single_step (tf2/run.py:557-622 on
# oracle/tfshim.py) -- the stand-in GradientTape cannot differentiate, so this is what pins the BACKWARD values (the oracle's torch
# autograd and the product's hand-written backward are both tested against it).  (model tag -> [(variable, directions)]);
# the differentiated loss follows the reference's gradient flow: the linear-eval head sits behind tf.stop_gradient (tf2/model.py:276-277),
# so encoder / projection-head variables see the contrastive loss only and the head's own variables the supervised loss + weight decay
GRAD_FD_VARS = {
    'r18_img': [('resnet/conv2d_fixed_padding/conv2d/kernel:0', 2),
                ('resnet/block_group3/residual_block_4/conv2d_fixed_padding_11/conv2d_11/kernel:0', 1),
                ('resnet/block_group3/residual_block_4/conv2d_fixed_padding_12/conv2d_12/kernel:0', 1),
                ('resnet/block_group3/residual_block_4/conv2d_fixed_padding_13/conv2d_13/kernel:0', 1),
                ('resnet/block_group3/residual_block_4/batch_norm_relu_13/sync_batch_normalization_13/gamma:0', 1),
                ('resnet/block_group3/residual_block_4/batch_norm_relu_13/sync_batch_normalization_13/beta:0', 1),
                ('projection_head/nl_0/dense/kernel:0', 1),
                ('projection_head/nl_2/batch_norm_relu_23/sync_batch_normalization_23/gamma:0', 1),
                ('head_supervised/linear_layer/dense_3/kernel:0', 1),
                ('head_supervised/linear_layer/dense_3/bias:0', 1)],
    # ResNet-50 (the small wiring case, every gamma / beta perturbed): the 7 x 7 stem, a strided projection shortcut and a strided 3 x 3 of
    # the first block of group 2, and all of an identity bottleneck block -- the block whose tail BatchNorm backward the product FOLDS into
    # its last convolution
    'r50': [('resnet/conv2d_fixed_padding/conv2d/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_3/conv2d_fixed_padding_11/conv2d_11/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_3/conv2d_fixed_padding_13/conv2d_13/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_4/conv2d_fixed_padding_15/conv2d_15/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_4/conv2d_fixed_padding_16/conv2d_16/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_4/conv2d_fixed_padding_17/conv2d_17/kernel:0', 1),
            ('resnet/block_group2/bottleneck_block_4/batch_norm_relu_17/sync_batch_normalization_17/gamma:0', 1),
            ('resnet/block_group2/bottleneck_block_4/batch_norm_relu_17/sync_batch_normalization_17/beta:0', 1)],
}
GRAD_FD_STEP = 1e-7


def grad_fd_directions(tag, shapes):
    """[(variable name, unit-norm direction of that variable's shape)]: deterministic by name (crc32-seeded normal draws).  DIRECTIONAL
    derivatives, not single coordinates: an fp32 forward flips a handful of ReLU signs that float64 does not, and one flip is a visible
    fraction of ONE weight coordinate's gradient (percent level) while <gradient, direction> over a whole tensor averages them out."""
    import zlib
    out = []
    for name, k in GRAD_FD_VARS[tag]:
        for j in range(k):
            d = np.random.default_rng([zlib.crc32(('%s#%d' % (name, j)).encode()), 5]).standard_normal(tuple(shapes[name]))
            out.append((name, d / np.sqrt((d * d).sum())))
    return out


MODEL_FLAGS = dict(width='width_multiplier', num_proj_layers='num_proj_layers', ft_proj_selector='ft_proj_selector',
                   proj_out_dim='proj_out_dim', global_bn='global_bn', lineareval='lineareval_while_pretraining')


# two-view augmentation (data_util.py:443-499): (source h, w, output size, strength, flip, jitter gate, grayscale gate)
AUG = [(61, 83, 32, 1.0, 1, 1, 0), (90, 70, 48, 1.0, 0, 1, 1), (64, 64, 40, 0.5, 1, 0, 1), (57, 120, 32, 1.0, 0, 0, 0), (100, 41, 24, 1.0, 1, 1, 1)]
AUG_EVAL = [(61, 83, 32), (90, 70, 48), (64, 64, 40), (33, 97, 24), (97, 33, 24)]
ASKED_KINDS = {'bbox': 0, 'uniform': 1, 'contrast': 2, 'saturation': 3, 'hue': 4, 'cropbox': 5}


def _rng(seed):
    return np.random.default_rng(1000 + seed)


def _aug_inputs(i):
    """source image (uint8) and the draws of one view: oracle/augment.py draws them, the gates are then forced per case"""
    from oracle import augment as oa
    sh, sw, size, strength, flip, jit, gray = AUG[i]
    rng = _rng(90 + i)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    p = oa.draw_train_params(rng, sh, sw, size, size, color_jitter_strength=strength)
    p[4], p[5], p[14] = flip, jit, gray
    return img, p


def _clip_box(box, shape):
    """a crop box drawn for another source size, moved inside this image (the two views of data.map_fn share ONE source image)"""
    y, x, h, w = (int(v) for v in box)
    h, w = min(h, shape[0]), min(w, shape[1])
    return np.array([min(y, shape[0] - h), min(x, shape[1] - w), h, w], dtype=np.float64)


def _asked_table(rows):
    out = np.zeros((len(rows), 12))
    for r, row in enumerate(rows):
        out[r, 0] = ASKED_KINDS[row[0]]
        out[r, 1:1 + len(row) - 1] = row[1:]
    return out


def _lars_inputs():
    rng = _rng(50)
    out = []
    for name, shp in LARS_NAMES:
        w = rng.standard_normal(shp) * 0.05
        g = rng.standard_normal(shp) * 1e-3
        if name.startswith('zero_weight'):
            w[:] = 0
        if name.startswith('zero_grad'):
            g[:] = 0
        out.append((name, w, g))
    return out


def _model_inputs(m):
    if m.get('recipe'):
        import recipe
        return recipe.structured_images(m['batch'], m['size'], 2, m['seed']), recipe.one_hot_labels(m['batch'], m['classes'], m['seed'])
    rng = _rng(70 + m['depth'] + int(m['sk'] > 0))
    images = rng.random((m['batch'], m['size'], m['size'], 6))
    labels = np.eye(m['classes'])[rng.integers(0, m['classes'], m['batch'])]
    return images, labels


def _oracle_model(m):
    import torch
    from oracle.model_torch import Config, init_model
    cfg = Config(resnet_depth=m['depth'], image_size=m['size'], sk_ratio=m['sk'], num_classes=m['classes'],
                 **{flag: m[k] for k, flag in MODEL_FLAGS.items() if k in m})
    params, state = init_model(cfg, seed=11, dtype=torch.float64)
    # the zero-initialised gammas (init_zero) and biases would hide wiring mistakes behind zeros: perturb every variable
    g = torch.Generator().manual_seed(12)
    init = {k: v.clone() for k, v in list(params.items()) + list(state.items())}
    if m.get('recipe'):
        import recipe
        for k in params:
            params[k] = torch.from_numpy(recipe.variable_value(k[len('model/'):], init[k].numpy(), m['perturb']))
        return cfg, params, state, init
    for k in params:
        if k.endswith('gamma:0'):
            params[k] = params[k] + 0.5 + torch.rand(params[k].shape, generator=g, dtype=torch.float64)
        elif k.endswith('beta:0') or k.endswith('bias:0'):
            params[k] = params[k] + 0.2 * torch.randn(params[k].shape, generator=g, dtype=torch.float64)
    return cfg, params, state, init


def _single_step(ref_dir, namespace):
    """tf2/run.py defines its training step as a function nested in main(); this compiles exactly that FunctionDef (line numbers kept)
    into `namespace`, which supplies what the closure supplied (model, optimizer, strategy, the metric objects, steps_per_loop)."""
    import ast
    path = os.path.join(ref_dir, 'tf2', 'run.py')
    tree = ast.parse(open(path).read(), filename=path)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'main')      # (perform_evaluation has a single_step too)
    node = next(n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n.name == 'single_step')
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)
    return namespace['single_step']


STEP_METRICS = ['contrast_loss', 'contrast_acc', 'contrast_entropy', 'supervised_loss', 'supervised_acc', 'weight_decay', 'total_loss']


# ---- reference side ---------------------------------------------------------------------------------------------------
def reference_cases(ref_dir=REFERENCE):
    from oracle import tfshim
    tf, FLAGS = tfshim.install()
    for k, v in FLAG_DEFAULTS.items():
        setattr(FLAGS, k, v)
    sys.path.insert(0, os.path.join(ref_dir, 'tf2'))
    try:
        for mname in ('objective', 'lars_optimizer', 'metrics', 'resnet', 'data_util', 'model'):
            sys.modules.pop(mname, None)
        objective, lars_optimizer, metrics, resnet, data_util, model = (
            importlib.import_module(n) for n in ('objective', 'lars_optimizer', 'metrics', 'resnet', 'data_util', 'model'))
    finally:
        sys.path.pop(0)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')         # 0 / 0 in the un-selected branch of LARS' tf.where (lars_optimizer.py:108-111)
        # objective.add_contrastive_loss, one replica (objective.py:35-89) + its gradient by central differences
        for i, c in enumerate(NTX):
            h = _rng(c['seed']).standard_normal((2 * c['n'], c['d']))
            f = lambda x: float(objective.add_contrastive_loss(tf.constant(x), hidden_norm=c['hidden_norm'], temperature=c['temperature'])[0])   # noqa: E731
            loss, logits_ab, labels = objective.add_contrastive_loss(tf.constant(h), hidden_norm=c['hidden_norm'], temperature=c['temperature'])
            g = np.zeros_like(h)
            for idx in np.ndindex(h.shape):
                hp, hm = h.copy(), h.copy()
                hp[idx] += 1e-6
                hm[idx] -= 1e-6
                g[idx] = (f(hp) - f(hm)) / 2e-6
            out['ntx%d_loss' % i], out['ntx%d_logits_ab' % i], out['ntx%d_labels' % i], out['ntx%d_grad_fd' % i] = (
                np.float64(loss), logits_ab.numpy(), labels.numpy(), g)
            acc, ent = tfshim._Mean(), tfshim._Mean()
            metrics.update_pretrain_metrics_train(tfshim._Mean(), acc, ent, loss, logits_ab, labels)       # metrics.py:22-35
            out['ntx%d_acc' % i], out['ntx%d_entropy' % i] = acc.result().numpy(), ent.result().numpy()
        # the strategy path (objective.py:58-68, 92-127) on R emulated replicas
        for i, c in enumerate(NTX_R):
            hs = [_rng(c['seed'] * 10 + r).standard_normal((2 * c['n'], c['d'])) for r in np.arange(c['R'])]
            strategy = tfshim.Strategy(c['R'])
            res = strategy.run(lambda x: objective.add_contrastive_loss(tf.constant(x), c['hidden_norm'], c['temperature'], strategy=strategy),
                               [(h,) for h in hs])
            cat = strategy.run(lambda x: objective.tpu_cross_replica_concat(tf.constant(x), strategy), [(h[:c['n']],) for h in hs])
            for r, (loss, logits_ab, labels) in enumerate(res):
                out['ntxr%d_%d_loss' % (i, r)], out['ntxr%d_%d_logits_ab' % (i, r)], out['ntxr%d_%d_labels' % (i, r)] = (
                    np.float64(loss), logits_ab.numpy(), labels.numpy())
            out['ntxr%d_concat' % i] = cat[0].numpy()
            assert all(np.array_equal(cat[0], c_) for c_ in cat)
        # objective.add_supervised_loss (objective.py:27-32) + metrics.update_finetune_metrics_train (metrics.py:48-55)
        rng = _rng(20)
        logits, lab = rng.standard_normal((9, 5)) * 3, np.eye(5)[rng.integers(0, 5, 9)]
        sl = objective.add_supervised_loss(tf.constant(lab), tf.constant(logits))
        acc = tfshim._Mean()
        metrics.update_finetune_metrics_train(tfshim._Mean(), acc, sl, tf.constant(lab), tf.constant(logits))
        out['sup_loss'], out['sup_acc'] = np.float64(sl), acc.result().numpy()
        # lars_optimizer.LARSOptimizer (lars_optimizer.py:83-157): momentum variants x name filters x zero norms, two steps
        for classic in (True, False):
            for nest in (False, True):
                vs = [tf.Variable(w, name=name) for name, w, _ in _lars_inputs()]
                opt = lars_optimizer.LARSOptimizer(0.3, momentum=0.9, use_nesterov=nest, weight_decay=1e-4, classic_momentum=classic,
                                                   exclude_from_weight_decay=LARS_EXCL)
                for step in (0, 1):
                    opt.apply_gradients([(g * (1 + step), v) for (_, _, g), v in zip(_lars_inputs(), vs)])
                    for j, v in enumerate(vs):
                        key = 'lars_c%d_n%d_s%d_v%d' % (classic, nest, step, j)
                        out[key + '_w'], out[key + '_m'] = v.numpy(), opt.get_slot(v, 'Momentum').numpy()
        # model.WarmUpAndCosineDecay / get_train_steps (model.py:72-110)
        for i, s in enumerate(SCHED):
            FLAGS.learning_rate_scaling, FLAGS.train_batch_size, FLAGS.warmup_epochs = s['scaling'], s['batch'], s['warmup_epochs']
            FLAGS.train_epochs, FLAGS.train_steps = s['train_epochs'], s['train_steps']
            sched = model.WarmUpAndCosineDecay(s['base_lr'], s['num_examples'])
            out['sched%d_lr' % i] = np.array([float(sched(st)) for st in SCHED_STEPS])
            out['sched%d_total_steps' % i] = np.int64(model.get_train_steps(s['num_examples']))
        for k in ('learning_rate_scaling', 'train_batch_size', 'warmup_epochs', 'train_epochs', 'train_steps'):
            setattr(FLAGS, k, FLAG_DEFAULTS[k])
        # data_util.gaussian_blur (data_util.py:323-361), the filter random_blur applies (kernel_size = height // 10)
        img = _rng(30).random((2, 40, 40, 3))
        out['blur'] = data_util.gaussian_blur(tf.constant(img), kernel_size=40 // 10, sigma=1.3, padding='SAME').numpy()
        out['blur_big'] = data_util.gaussian_blur(tf.constant(img), kernel_size=9, sigma=0.4, padding='SAME').numpy()
        # data_util.preprocess_image (data_util.py:443-520) with SCRIPTED draws: the reference's composition -- crop -> resize -> flip ->
        # [0.8] colour jitter in shuffled order with a clip after every op -> [0.2] grayscale -> clip -- and the ranges it draws from;
        # the pixel kernels underneath are oracle/augment.py's (tfshim.py, "tf.random / tf.image (scripted)")
        for i in np.arange(len(AUG)):
            img, p = _aug_inputs(int(i))
            size, strength = AUG[i][2], AUG[i][3]
            tfshim.SCRIPT.clear()
            tfshim.SCRIPT.update(crop=p[0:4], flip=bool(p[4]), perm=p[6:10], contrast=p[11], saturation=p[12], hue=p[13],
                                 uniform=[p[10]], unit=[0.5, 0.5, 0.1 if p[5] else 0.9, 0.1 if p[14] else 0.9], asked=[])
            y = data_util.preprocess_image(img, size, size, is_training=True, color_jitter_strength=strength)
            out['aug%d_params' % i], out['aug%d_out' % i] = p, y.numpy()
            out['aug%d_asked' % i] = _asked_table(tfshim.SCRIPT['asked'])
            assert not tfshim.SCRIPT['unit'], 'a gate of random_apply was not drawn'
        # data.map_fn (data.py:52-62, nested in build_input_fn: compiled from its own source like run.py's step): two transformations
        # of one image concatenated on the channel axis + the one-hot label, through data.get_preprocess_fn (data.py:101-117)
        sys.modules.setdefault('tensorflow_datasets', importlib.util.module_from_spec(importlib.machinery.ModuleSpec('tensorflow_datasets', None)))
        sys.path.insert(0, os.path.join(ref_dir, 'tf2'))
        try:
            sys.modules.pop('data', None)
            data = importlib.import_module('data')
        finally:
            sys.path.pop(0)
        import ast
        dpath = os.path.join(ref_dir, 'tf2', 'data.py')
        dnode = next(n for n in ast.walk(ast.parse(open(dpath).read(), filename=dpath)) if isinstance(n, ast.FunctionDef) and n.name == 'map_fn')
        for i in (0, 1):
            views = [_aug_inputs(i), _aug_inputs(i + 2)]          # the draws of view 0 / view 1 (the source image is view 0's)
            img = views[0][0]
            FLAGS.image_size, FLAGS.color_jitter_strength = AUG[i][2], AUG[i][3]
            ref_fn = data.get_preprocess_fn(True, is_pretrain=True)
            calls = []

            def scripted(image, _fn=ref_fn, _views=views, _calls=calls):
                p = _views[len(_calls)][1]
                _calls.append(1)
                tfshim.SCRIPT.clear()
                tfshim.SCRIPT.update(crop=_clip_box(p[0:4], image.shape), flip=bool(p[4]), perm=p[6:10], contrast=p[11], saturation=p[12],
                                     hue=p[13], uniform=[p[10]], unit=[0.5, 0.5, 0.1 if p[5] else 0.9, 0.1 if p[14] else 0.9], asked=[])
                return _fn(image)
            ns = dict(tf=tf, FLAGS=FLAGS, is_training=True, preprocess_fn_pretrain=scripted, preprocess_fn_finetune=None, num_classes=7)
            exec(compile(ast.Module(body=[dnode], type_ignores=[]), dpath, 'exec'), ns)
            image6, label = ns['map_fn'](img, 3)
            out['twoview%d_image' % i], out['twoview%d_label' % i] = image6.numpy(), label.numpy()
        FLAGS.image_size = FLAG_DEFAULTS['image_size']
        for i, (sh, sw, size) in enumerate(AUG_EVAL):
            img = _rng(95 + i).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
            tfshim.SCRIPT.clear()
            tfshim.SCRIPT.update(asked=[])
            out['augeval%d_out' % i] = data_util.preprocess_image(img, size, size, is_training=False, test_crop=True).numpy()
            out['augeval%d_box' % i] = _asked_table(tfshim.SCRIPT['asked'])        # the centre-crop box of _compute_crop_shape (:175-243)
        # resnet.Conv2dFixedPadding / FixedPadding (resnet.py:160-208)
        for (hh, k, s) in CONV_CASES:
            tfshim.reset_uids()
            x = _rng(40 + hh + k).standard_normal((2, hh, hh, 4))
            w = _rng(41 + hh + k).standard_normal((k, k, 4, 6))
            layer = resnet.Conv2dFixedPadding(filters=6, kernel_size=k, strides=s)
            layer(tf.constant(x), training=True)
            layer.conv2d.kernel.assign(w)
            out['conv_h%d_k%d_s%d' % (hh, k, s)] = layer(tf.constant(x), training=True).numpy()
        # resnet.BatchNormRelu (resnet.py:31-78), local and "global" (one replica) statistics, training + inference
        for gbn in (False, True):
            FLAGS.global_bn = gbn
            tfshim.reset_uids()
            x = _rng(60).standard_normal((6, 5, 4, 8)) * 2 + 0.5
            layer = resnet.BatchNormRelu(relu=True)
            out['bn_g%d_train' % gbn] = layer(tf.constant(x), training=True).numpy()
            out['bn_g%d_moving_mean' % gbn], out['bn_g%d_moving_var' % gbn] = layer.bn.moving_mean.numpy(), layer.bn.moving_variance.numpy()
            out['bn_g%d_eval' % gbn] = layer(tf.constant(x), training=False).numpy()
            out['bn_g%d_names' % gbn] = np.array(sorted(v.name for v in layer.variables))
        FLAGS.global_bn = True
        # model.Model (model.py:224-280) over resnet.resnet() (resnet.py:529-747): names, shapes, initial values, forward outputs,
        # moving statistics, weight decay -- with the oracle's (perturbed) variables injected by NAME
        FLAGS.use_blur = False
        for m in MODELS:
            FLAGS.resnet_depth, FLAGS.image_size, FLAGS.sk_ratio = m['depth'], m['size'], m['sk']
            for k, flag in MODEL_FLAGS.items():
                setattr(FLAGS, flag, m.get(k, FLAG_DEFAULTS[flag]))
            tfshim.reset_uids()
            net = model.Model(m['classes'])
            images, labels = _model_inputs(m)
            net(tf.constant(images), training=False)                                  # builds the variables (inference: no moving-average update)
            vs = sorted(tfshim.CREATED_VARIABLES, key=lambda v: v.name)      # per-variable tables are sorted by name on both sides
            t = m['tag']
            out[t + '_names'] = np.array([v.name for v in vs])
            out[t + '_shapes'] = np.array([' '.join(str(d) for d in v.value.shape) for v in vs])
            out[t + '_trainable'] = np.array([v.trainable for v in vs])
            out[t + '_init_checksum'] = np.array([[float(v.numpy().sum()), float(np.abs(v.numpy()).sum())] if not v.name.endswith('kernel:0')
                                                  else [0.0, 0.0] for v in vs])    # kernels are random in the reference: not compared
            cfg, params, state, _ = _oracle_model(m)
            allv = {**params, **state}
            for v in vs:
                v.assign(allv['model/' + v.name].numpy())
            proj, sup = net(tf.constant(images), training=True)
            none = np.zeros((0,))
            out[t + '_proj'], out[t + '_sup'] = proj.numpy(), sup.numpy() if sup is not None else none
            out[t + '_moving_checksum'] = np.array([[float(v.numpy().sum()), float(np.abs(v.numpy()).sum())] for v in vs
                                                    if 'moving_' in v.name])
            proj_e, sup_e = net(tf.constant(images), training=False)
            out[t + '_proj_eval'], out[t + '_sup_eval'] = proj_e.numpy(), sup_e.numpy() if sup_e is not None else none
            FLAGS.weight_decay = 1e-4
            out[t + '_wd_lars'] = np.float64(model.add_weight_decay(net, adjust_per_optimizer=True))       # model.py:47-60
            out[t + '_wd_all'] = np.float64(model.add_weight_decay(net, adjust_per_optimizer=False))       # model.py:62-69
            # which variables model.build_optimizer's LARSOptimizer decays / adapts (lars_optimizer.py:139-157 on the reference's own names)
            FLAGS.optimizer, FLAGS.momentum = 'lars', 0.9
            lopt = model.build_optimizer(0.1)
            tv = sorted(v.name for v in net.trainable_variables)
            out[t + '_lars_names'] = np.array(tv)
            out[t + '_lars_decays'] = np.array([bool(lopt._use_weight_decay(n)) for n in tv])
            out[t + '_lars_adapts'] = np.array([bool(lopt._do_layer_adaptation(n)) for n in tv])
            FLAGS.weight_decay = FLAG_DEFAULTS['weight_decay']
        # run.py:557-622 single_step itself: loss composition, the division by the replica count, the metric updates, the variables handed
        # to the optimizer -- on one replica and on two emulated replicas (SyncBatchNormalization + cross-replica concat in lock step)
        for tag, R in STEPS:
            m = next(mm for mm in MODELS if mm['tag'] == tag)
            FLAGS.resnet_depth, FLAGS.image_size, FLAGS.sk_ratio = m['depth'], m['size'], m['sk']
            for k, flag in MODEL_FLAGS.items():
                setattr(FLAGS, flag, m.get(k, FLAG_DEFAULTS[flag]))
            FLAGS.weight_decay = 1e-4
            tfshim.reset_uids()
            net = model.Model(m['classes'])
            images, labels = _model_inputs(m)
            net(tf.constant(images), training=False)
            cfg, params, state, _ = _oracle_model(m)
            allv = {**params, **state}
            for v in tfshim.CREATED_VARIABLES:
                v.assign(allv['model/' + v.name].numpy())
            strategy = tfshim.Strategy(R)
            opt = tfshim.RecordingOptimizer()
            mets = {k: tfshim._Mean('train/' + k) for k in STEP_METRICS}
            ns = dict(tf=tf, FLAGS=FLAGS, logging=sys.modules['absl.logging'], metrics=metrics, obj_lib=objective, model_lib=model,
                      optimizer=opt, steps_per_loop=100, model=net, strategy=strategy, contrast_loss_metric=mets['contrast_loss'],
                      contrast_acc_metric=mets['contrast_acc'], contrast_entropy_metric=mets['contrast_entropy'],
                      supervised_loss_metric=mets['supervised_loss'], supervised_acc_metric=mets['supervised_acc'],
                      weight_decay_metric=mets['weight_decay'], total_loss_metric=mets['total_loss'])
            step = _single_step(ref_dir, ns)
            per = m['batch'] // R
            assert per * R == m['batch'] or R == 1
            shards = [(tf.constant(images[r * per:(r + 1) * per] if R > 1 else images),
                       {'labels': tf.constant(labels[r * per:(r + 1) * per] if R > 1 else labels)}) for r in np.arange(R)]
            tfshim.GradientTape.recorded.clear()
            strategy.run(step, shards)
            key = 'step_%s_R%d' % (tag, R)
            out[key + '_scaled_loss'] = np.array([tfshim.GradientTape.recorded[r][0] for r in np.arange(R)])
            out[key + '_metrics'] = np.array([float(mets[k].result()) for k in STEP_METRICS])
            out[key + '_applied_names'] = np.array(sorted(opt.applied[0]))
            assert all(a == opt.applied[0] for a in opt.applied) and len(opt.applied) == R
            assert tfshim.GradientTape.recorded[0][1] == opt.applied[0]
            if tag in GRAD_FD_VARS and R == 1:
                byname = {v.name: v for v in tfshim.CREATED_VARIABLES}

                def losses():
                    fresh = {k: tfshim._Mean('train/' + k) for k in STEP_METRICS}
                    for k in STEP_METRICS:
                        ns[k + '_metric'] = fresh[k]          # single_step reads its metric objects from this namespace at call time
                    strategy.run(step, shards)
                    return {k: float(fresh[k].result()) for k in STEP_METRICS}
                g = []
                for name, d in grad_fd_directions(tag, {n: v.value.shape for n, v in byname.items()}):
                    v = byname[name]
                    base = v.numpy().copy()
                    vals = []
                    for sgn in (+1.0, -1.0):
                        v.assign(base + sgn * GRAD_FD_STEP * d)
                        m_ = losses()
                        # the loss this variable's gradient flows from (stop_gradient in front of the linear-eval head, model.py:276-277)
                        vals.append(m_['supervised_loss'] + m_['weight_decay'] if 'head_supervised' in name else m_['contrast_loss'])
                    v.assign(base)
                    g.append((vals[0] - vals[1]) / (2.0 * GRAD_FD_STEP))
                out[key + '_grad_fd'] = np.array(g)
        FLAGS.weight_decay = FLAG_DEFAULTS['weight_decay']
        for k in ('resnet_depth', 'image_size', 'sk_ratio', 'use_blur') + tuple(MODEL_FLAGS.values()):
            setattr(FLAGS, k, FLAG_DEFAULTS[k])
    return out


# ---- oracle side ------------------------------------------------------------------------------------------------------
def oracle_cases():
    import torch
    from oracle import blur as oblur
    from oracle import lars as olars
    from oracle import ntxent as ont
    from oracle.model_torch import Builder, Config
    out = {}
    for i, c in enumerate(NTX):
        h = _rng(c['seed']).standard_normal((2 * c['n'], c['d']))
        loss, logits_ab, labels = ont.add_contrastive_loss(h, c['hidden_norm'], c['temperature'])
        _, grads = ont.contrastive_loss_and_grad([h], c['hidden_norm'], c['temperature'])
        acc, ent = ont.contrastive_metrics(logits_ab, labels)
        out['ntx%d_loss' % i], out['ntx%d_logits_ab' % i], out['ntx%d_labels' % i], out['ntx%d_grad_fd' % i] = (
            np.float64(loss), logits_ab, labels, grads[0])
        out['ntx%d_acc' % i], out['ntx%d_entropy' % i] = np.float64(acc), np.float64(ent)
    for i, c in enumerate(NTX_R):
        hs = [_rng(c['seed'] * 10 + r).standard_normal((2 * c['n'], c['d'])) for r in range(c['R'])]
        for r in range(c['R']):
            loss, logits_ab, labels = ont.add_contrastive_loss(hs[r], c['hidden_norm'], c['temperature'], all_hiddens=hs, replica_id=r)
            out['ntxr%d_%d_loss' % (i, r)], out['ntxr%d_%d_logits_ab' % (i, r)], out['ntxr%d_%d_labels' % (i, r)] = (
                np.float64(loss), logits_ab, labels)
        out['ntxr%d_concat' % i] = ont.tpu_cross_replica_concat([h[:c['n']] for h in hs])
    rng = _rng(20)
    logits, lab = rng.standard_normal((9, 5)) * 3, np.eye(5)[rng.integers(0, 5, 9)]
    lt = torch.from_numpy(logits)
    out['sup_loss'] = np.float64(-(torch.from_numpy(lab) * torch.log_softmax(lt, 1)).sum(1).mean())     # as model_torch.single_step_losses
    out['sup_acc'] = np.float64(np.mean(lab.argmax(1) == logits.argmax(1)))
    for classic in (True, False):
        for nest in (False, True):
            ws = [w.copy() for _, w, _ in _lars_inputs()]
            ms = [np.zeros_like(w) for w in ws]
            for step in (0, 1):
                for j, (name, _, g) in enumerate(_lars_inputs()):
                    ws[j], ms[j] = olars.lars_apply(name + ':0', ws[j], g * (1 + step), ms[j], 0.3, momentum=0.9, use_nesterov=nest,
                                                    weight_decay=1e-4, classic_momentum=classic, exclude_from_weight_decay=LARS_EXCL)
                    key = 'lars_c%d_n%d_s%d_v%d' % (classic, nest, step, j)
                    out[key + '_w'], out[key + '_m'] = ws[j], ms[j]
    for i, s in enumerate(SCHED):
        kw = dict(warmup_epochs=s['warmup_epochs'], train_batch_size=s['batch'], learning_rate_scaling=s['scaling'],
                  train_epochs=s['train_epochs'], train_steps=s['train_steps'])
        out['sched%d_lr' % i] = np.array([olars.warmup_and_cosine_decay(st, s['base_lr'], s['num_examples'], **kw) for st in SCHED_STEPS])
        out['sched%d_total_steps' % i] = np.int64(olars.get_train_steps(s['num_examples'], s['train_steps'], s['train_epochs'], s['batch']))
    img = _rng(30).random((2, 40, 40, 3))
    out['blur'] = oblur.gaussian_blur(img, 40 // 10, 1.3)
    out['blur_big'] = oblur.gaussian_blur(img, 9, 0.4)
    from oracle import augment as oa
    for i in range(len(AUG)):
        img, p = _aug_inputs(i)
        size, strength = AUG[i][2], AUG[i][3]
        out['aug%d_params' % i], out['aug%d_out' % i] = p, oa.apply_train_params(img, p, size, size)
        # what oracle/augment.py::draw_train_params draws from (augment.py:225-247), in the order the reference asks
        ar = size / size
        rows = [('bbox', 0.1, 3. / 4 * ar, 4. / 3. * ar, 0.08, 1.0, 100, 1.0, 0.0, 0.0, 1.0, 1.0), ('cropbox',) + tuple(p[0:4])]
        if p[5] > 0:
            b_, c_, s_, h_ = 0.8 * strength, 0.8 * strength, 0.8 * strength, 0.2 * strength
            ops = {0: ('uniform', max(1.0 - b_, 0.0), 1.0 + b_), 1: ('contrast', 1 - c_, 1 + c_), 2: ('saturation', 1 - s_, 1 + s_), 3: ('hue', -h_, h_)}
            rows += [ops[int(v)] for v in p[6:10]]
        out['aug%d_asked' % i] = _asked_table(rows)
    for i in (0, 1):
        views = [_aug_inputs(i), _aug_inputs(i + 2)]
        img = views[0][0]
        params = np.stack([np.concatenate([_clip_box(v[1][0:4], img.shape), v[1][4:]]) for v in views])[None]
        out['twoview%d_image' % i] = oa.two_view_batch([img], params, AUG[i][2], AUG[i][2])[0]
        out['twoview%d_label' % i] = np.eye(7)[3]
    for i, (sh, sw, size) in enumerate(AUG_EVAL):
        img = _rng(95 + i).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        out['augeval%d_out' % i] = oa.preprocess_for_eval(img, size, size)
        out['augeval%d_box' % i] = _asked_table([('cropbox',) + tuple(oa.center_crop_box(sh, sw, size, size, 0.875))])
    for (hh, k, s) in CONV_CASES:
        x = _rng(40 + hh + k).standard_normal((2, hh, hh, 4))
        w = _rng(41 + hh + k).standard_normal((k, k, 4, 6))
        b = Builder(Config(), dtype=torch.float64)
        b.init = False
        b.params = {'conv2d_fixed_padding/conv2d/kernel:0': torch.from_numpy(w)}
        out['conv_h%d_k%d_s%d' % (hh, k, s)] = b.conv2d_fixed_padding(torch.from_numpy(x).permute(0, 3, 1, 2), 6, k, s).permute(0, 2, 3, 1).numpy()
    for gbn in (False, True):
        x = _rng(60).standard_normal((6, 5, 4, 8)) * 2 + 0.5
        xt = torch.from_numpy(x).permute(0, 3, 1, 2)
        b = Builder(Config(global_bn=gbn), dtype=torch.float64)
        out['bn_g%d_train' % gbn] = b.batch_norm_relu(xt).permute(0, 2, 3, 1).numpy()
        st = {k.rsplit('/', 1)[1]: v.numpy() for k, v in b.new_state.items()}
        out['bn_g%d_moving_mean' % gbn], out['bn_g%d_moving_var' % gbn] = st['moving_mean:0'], st['moving_variance:0']
        b2 = Builder(Config(global_bn=gbn), params=b.params, dtype=torch.float64, state={k: v.double() for k, v in b.new_state.items()})
        b2.training = False
        out['bn_g%d_eval' % gbn] = b2.batch_norm_relu(xt).permute(0, 2, 3, 1).numpy()
        out['bn_g%d_names' % gbn] = np.array(sorted(list(b.params.keys()) + list(b.state.keys())))
    for m in MODELS:
        cfg, params, state, init = _oracle_model(m)
        images, labels = _model_inputs(m)
        t = m['tag']
        names = sorted(list(params.keys()) + list(state.keys()))
        out[t + '_names'] = np.array([n[len('model/'):] for n in names])
        out[t + '_shapes'] = np.array([' '.join(str(d) for d in init[n].shape) for n in names])
        out[t + '_trainable'] = np.array([n in params for n in names])
        out[t + '_init_checksum'] = np.array([[float(init[n].sum()), float(init[n].abs().sum())] if not n.endswith('kernel:0') else [0.0, 0.0]
                                              for n in names])
        b = Builder(cfg, params=params, state=state, dtype=torch.float64)
        with torch.no_grad():
            proj, sup = b.model(torch.from_numpy(images), training=True)
        none = np.zeros((0,))
        out[t + '_proj'], out[t + '_sup'] = proj.numpy(), sup.numpy() if sup is not None else none
        out[t + '_moving_checksum'] = np.array([[float(b.new_state[n].sum()), float(b.new_state[n].abs().sum())] for n in names if 'moving_' in n])
        b2 = Builder(cfg, params=params, state={k: v for k, v in b.new_state.items()}, dtype=torch.float64)
        with torch.no_grad():
            proj_e, sup_e = b2.model(torch.from_numpy(images), training=False)
        out[t + '_proj_eval'], out[t + '_sup_eval'] = proj_e.numpy(), sup_e.numpy() if sup_e is not None else none
        out[t + '_wd_lars'] = np.float64(olars.add_weight_decay_lars([(n, p.numpy()) for n, p in params.items()], 1e-4))
        out[t + '_wd_all'] = np.float64(1e-4 * sum(0.5 * float((p * p).sum()) for n, p in params.items() if 'batch_normalization' not in n))
        tv = sorted(n[len('model/'):] for n in params)
        out[t + '_lars_names'] = np.array(tv)
        out[t + '_lars_decays'] = np.array([olars.use_weight_decay(n, 1e-4, LARS_EXCL) for n in tv])
        out[t + '_lars_adapts'] = np.array([olars.do_layer_adaptation(n, LARS_EXCL) for n in tv])
    # run.py:557-622: R replicas with SyncBatchNormalization, the differentiable concat and loss / R are ONE replica on the global batch
    # (tests/test_oracle.py::test_sharded_equals_global_batch) -- the oracle's single_step_losses on the whole batch gives every number
    import dataclasses
    from oracle.model_torch import single_step_losses
    for tag, R in STEPS:
        m = next(mm for mm in MODELS if mm['tag'] == tag)
        cfg, params, state, _ = _oracle_model(m)
        cfg = dataclasses.replace(cfg, weight_decay=1e-4)
        images, labels = _model_inputs(m)
        if R > 1:
            per = m['batch'] // R
            images, labels = images[:per * R], labels[:per * R]
        with torch.no_grad():
            o = single_step_losses(cfg, params, state, torch.from_numpy(images), torch.from_numpy(labels))
        acc, ent = ont.contrastive_metrics(o['logits_con'].numpy(), o['labels_con'].numpy())
        l2 = np.concatenate([labels, labels], 0)
        sup_acc = float(np.mean(l2.argmax(1) == o['sup_logits'].numpy().argmax(1)))
        key = 'step_%s_R%d' % (tag, R)
        # replica r's own loss (objective.py:58-87 with the gathered hiddens; its rows of the linear-eval logits), divided by R
        N = images.shape[0]
        per = N // R
        proj, sup = o['proj'].numpy(), o['sup_logits'].numpy()
        rows = [np.concatenate([np.arange(r * per, (r + 1) * per), N + np.arange(r * per, (r + 1) * per)]) for r in range(R)]
        hs = [proj[ix] for ix in rows]
        scaled = []
        for r in range(R):
            con_r, _, _ = ont.add_contrastive_loss(hs[r], cfg.hidden_norm, cfg.temperature, all_hiddens=hs if R > 1 else None, replica_id=r)
            lr_ = np.concatenate([labels[r * per:(r + 1) * per]] * 2, 0)
            z = sup[rows[r]]
            zmax = z.max(1, keepdims=True)
            sup_r = float(np.mean(np.log(np.exp(z - zmax).sum(1)) + zmax[:, 0] - (lr_ * z).sum(1)))
            scaled.append((float(con_r) + sup_r + float(o['weight_decay'])) / R)
        out[key + '_scaled_loss'] = np.array(scaled)
        assert abs(sum(scaled) - float(o['total_loss'])) <= 1e-9 * abs(float(o['total_loss']))      # sharded == global batch
        out[key + '_metrics'] = np.array([float(o['con_loss']), acc, ent, float(o['sup_loss']), sup_acc, float(o['weight_decay']),
                                          float(o['total_loss'])])
        out[key + '_applied_names'] = np.array(sorted(n[len('model/'):] for n in params))
        if tag in GRAD_FD_VARS and R == 1:
            # the same directions from torch autograd of the oracle's step (train_step = what tape.gradient would return)
            from collections import OrderedDict
            from oracle.model_torch import train_step
            mom = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
            _, _, _, t = train_step(cfg, params, state, mom, torch.from_numpy(images), torch.from_numpy(labels), 0.1)
            dirs = grad_fd_directions(tag, {n[len('model/'):]: tuple(v.shape) for n, v in params.items()})
            out[key + '_grad_fd'] = np.array([float((t['grads']['model/' + n].double().numpy() * d).sum()) for n, d in dirs])
    return out


def compare(a, b, key):
    """relative max-abs error of two fixture entries (0 / 1 for string and boolean tables); tolerance class by key"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return float('inf')
    if a.dtype.kind in 'US' or a.dtype == bool:
        return 0.0 if bool(np.all(a == b)) else 1.0
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / (np.max(np.abs(a.astype(np.float64))) + 1e-300))


def tolerance(key):
    if key.endswith('_grad_fd'):
        return 1e-6 if key.startswith('ntx') else 2e-5     # central differences of the reference loss in float64 (step 1e-6 / 1e-5 through a ReLU network)
    if key.endswith('_acc'):
        return 1e-6          # oracle/ntxent.py averages the hits in float32 as metrics.py:30 casts them
    return 1e-8              # float64 on both sides; ~50-layer forward passes agree to 1e-10


if __name__ == '__main__':
    if '--check' in sys.argv:
        # regenerate from the reference source and compare with the committed fixtures (exit 1 on any difference)
        ref = reference_cases()
        old = dict(np.load(OUT_NPZ))
        bad = [k for k in sorted(set(ref) | set(old)) if k not in ref or k not in old or compare(old[k], ref[k], k) > 1e-12]
        print('%d fixtures regenerated from %s, %d differ%s' % (len(ref), REFERENCE, len(bad), (': ' + ', '.join(bad[:8])) if bad else ''))
        sys.exit(1 if bad else 0)
    ref = reference_cases()
    np.savez_compressed(OUT_NPZ, **ref)
    meta = dict(reference=REFERENCE, files=['tf2/objective.py', 'tf2/lars_optimizer.py', 'tf2/metrics.py', 'tf2/resnet.py', 'tf2/data_util.py',
                                            'tf2/model.py'], executed_on='oracle/tfshim.py (float64 numpy stand-in for the TensorFlow calls)',
                keys=len(ref), flags=FLAG_DEFAULTS)
    json.dump(meta, open(OUT_JSON, 'w'), indent=1)
    print('wrote %d arrays to %s (%.1f KB)' % (len(ref), OUT_NPZ, os.path.getsize(OUT_NPZ) / 1024))
