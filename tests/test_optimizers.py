"""build_optimizer's three branches (/root/reference/tf2/model.py:29-44) and the CPU restatement of the two Keras optimizers
(oracle/optimizers.py) against hand-derived numbers; the HIP multi-tensor kernels against that restatement on the GPU."""
import numpy as np
import pytest


def test_oracle_sgd_and_adam_hand_derived():
    """Paper-and-pencil cases (not oracle-generated).  SGD, lr 0.1, momentum 0.9, nesterov, w = 1, g = 0.5, accum = 0.2:
    accum' = 0.9 * 0.2 - 0.1 * 0.5 = 0.13; w' = 1 + 0.9 * 0.13 - 0.05 = 1.067.  Without nesterov: w' = 1.13.
    Adam, t = 1, lr 0.001, g = 0.5, m = v = 0: m' = 0.05, v' = 0.00025; lr_t = 0.001 * sqrt(0.001) / 0.1;
    w' = 1 - lr_t * 0.05 / (sqrt(0.00025) + 1e-7) = 1 - 0.001 * (1 / (1 + 1e-7 / 0.0158113883...))."""
    from oracle import optimizers as oo
    w, a = oo.sgd_apply([1.0], [0.5], [0.2], 0.1, 0.9, True)
    assert abs(a[0] - 0.13) < 1e-15 and abs(w[0] - 1.067) < 1e-15
    w, a = oo.sgd_apply([1.0], [0.5], [0.2], 0.1, 0.9, False)
    assert abs(w[0] - 1.13) < 1e-15
    w, a = oo.sgd_apply([2.0], [0.5], [0.0], 0.1, 0.0, False, l2=0.25)      # g + l2 w = 1.0
    assert abs(w[0] - 1.9) < 1e-15
    w, m, v = oo.adam_apply([1.0], [0.5], [0.0], [0.0], 0.001, 1)
    # (1 - beta) is formed in float32, as in the TensorFlow kernel: 1 - 0.9f = 0.100000024, 1 - 0.999f = 0.00099998713
    assert abs(m[0] - 0.05) < 2e-8 and abs(v[0] - 0.00025) < 2e-8 and abs(v[0] / 0.00025 - 1 + 1.2875e-5) < 1e-7
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
    want = 1.0 - 0.001 * (np.sqrt(1 - b2) / (1 - b1)) * m[0] / (np.sqrt(v[0]) + 1e-7)
    assert abs(w[0] - want) < 1e-15 and abs(w[0] - (1.0 - 0.001)) < 1e-7          # the first Adam step moves by ~lr
    # second step: the bias correction moves with t
    w2, m2, v2 = oo.adam_apply(w, [0.25], m, v, 0.001, 2)
    lr_t = 0.001 * np.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    assert abs(m2[0] - (0.9 * 0.05 + 0.1 * 0.25)) < 1e-7
    assert abs(w2[0] - (w[0] - lr_t * m2[0] / (np.sqrt(v2[0]) + 1e-7))) < 1e-15


def test_build_optimizer_branches():
    """tf2/model.py:29-44: 'momentum' -> SGD(lr, FLAGS.momentum, nesterov=True), 'adam' -> Adam(lr), 'lars' -> LARSOptimizer with the
    reference's name filters, anything else -> ValueError('Unknown optimizer ...')."""
    from simclr_amd import model as model_lib
    from simclr_amd import optimizers
    from simclr_amd.flags import FLAGS
    from simclr_amd.lars_optimizer import LARSOptimizer
    FLAGS.reset()
    try:
        FLAGS.update(optimizer='momentum', momentum=0.8, weight_decay=1e-4)
        o = model_lib.build_optimizer(0.3)
        assert isinstance(o, optimizers.SGD) and o.momentum == 0.8 and o.nesterov is True and o.current_lr() == 0.3 and o.l2 == 1e-4
        assert o._takes_l2('conv2d/kernel:0') and not o._takes_l2('sync_batch_normalization_3/gamma:0')
        FLAGS.update(optimizer='adam')
        o = model_lib.build_optimizer(lambda step: 0.01 * (step + 1))
        assert isinstance(o, optimizers.Adam) and (o.beta_1, o.beta_2, o.epsilon) == (0.9, 0.999, 1e-7) and o.current_lr() == 0.01
        FLAGS.update(optimizer='lars')
        o = model_lib.build_optimizer(0.1)
        assert isinstance(o, LARSOptimizer) and o.exclude_from_weight_decay == ['batch_normalization', 'bias', 'head_supervised']
        FLAGS.update(optimizer='rmsprop')
        with pytest.raises(ValueError, match='Unknown optimizer'):
            model_lib.build_optimizer(0.1)
    finally:
        FLAGS.reset()


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['sgd_nesterov', 'sgd_plain', 'adam'])
def test_multi_tensor_sgd_adam_kernels_vs_oracle(kind):
    """simclr_sgd_multi_tensor / simclr_adam_multi_tensor: three steps over tensors that span several 8192-element chunks, a
    BatchNorm-named tensor (no l2 term) and a zero tensor, against oracle/optimizers.py in float64."""
    import torch
    from oracle import optimizers as oo
    from simclr_amd import optimizers
    from simclr_amd.lars_optimizer import Variable
    g = np.random.default_rng(5)
    shapes = [('conv2d/kernel:0', (3, 3, 64, 64)), ('sync_batch_normalization/gamma:0', (64,)), ('dense/kernel:0', (300, 77)),
              ('head_supervised/linear_layer/dense/bias:0', (10,)), ('zero/kernel:0', (1000,)), ('conv2d_1/kernel:0', (1, 1, 256, 1030))]
    vs, ref = [], {}
    for name, shp in shapes:
        w = (g.standard_normal(shp) * 0.05).astype(np.float32)
        if name.startswith('zero'):
            w[:] = 0
        v = Variable(name, torch.from_numpy(w).cuda())
        v.grad = torch.zeros_like(v.value)
        vs.append(v)
        ref[name] = dict(w=w.astype(np.float64), s=[np.zeros(shp), np.zeros(shp)])
    lr, l2 = 0.05, 1e-3
    if kind == 'adam':
        opt = optimizers.Adam(lr, l2=l2)
    else:
        opt = optimizers.SGD(lr, 0.9, nesterov=(kind == 'sgd_nesterov'), l2=l2)
    for step in range(3):
        for v in vs:
            gr = (g.standard_normal(v.shape) * 1e-2).astype(np.float32)
            v.grad.copy_(torch.from_numpy(gr).cuda())
            r = ref[v.name]
            c = 0.0 if 'batch_normalization' in v.name else l2
            if kind == 'adam':
                r['w'], r['s'][0], r['s'][1] = oo.adam_apply(r['w'], gr, r['s'][0], r['s'][1], lr, step + 1, l2=c)
            else:
                r['w'], r['s'][0] = oo.sgd_apply(r['w'], gr, r['s'][0], lr, 0.9, kind == 'sgd_nesterov', l2=c)
        opt.apply_gradients([(v.grad, v) for v in vs])
    torch.cuda.synchronize()
    assert opt.iterations == 3
    for v in vs:
        r = ref[v.name]
        got = v.value.double().cpu().numpy()
        assert np.abs(got - r['w']).max() <= 4e-6 * max(np.abs(r['w']).max(), 1e-3) + 1e-9, v.name
        if kind == 'adam':
            assert np.abs(opt.get_slot(v, 'm').double().cpu().numpy() - r['s'][0]).max() <= 4e-6 * max(np.abs(r['s'][0]).max(), 1e-6)
            assert np.abs(opt.get_slot(v, 'v').double().cpu().numpy() - r['s'][1]).max() <= 4e-6 * max(np.abs(r['s'][1]).max(), 1e-9) + 1e-10
        else:
            assert np.abs(opt.get_slot(v).double().cpu().numpy() - r['s'][0]).max() <= 4e-6 * max(np.abs(r['s'][0]).max(), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('optimizer', ['momentum', 'adam'])
def test_train_step_with_the_other_optimizers(optimizer):
    """One ResNet-18 / 32 px step with --optimizer=momentum | adam: the step runs, the loss is finite, every trainable variable moved,
    and the linear-eval head's kernel (which takes the weight-decay term, tf2/model.py:62-69) moved by what the oracle rule gives from
    the step's own gradient."""
    import torch
    from oracle import optimizers as oo
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step, synthetic_batches
    FLAGS.reset()
    try:
        FLAGS.update(resnet_depth=18, image_size=32, train_batch_size=16, compute_dtype='f32', use_blur=False, optimizer=optimizer,
                     weight_decay=1e-3, momentum=0.9)
        RT.reset()
        RT.device = torch.device('cuda')
        model = model_lib.Model(10)
        opt = model_lib.build_optimizer(0.01)
        step = make_single_step(model, opt, None)
        data = synthetic_batches(16, 32, 10, RT.device, seed=0)
        f, l = next(data)
        model(f, training=False)
        before = {v.name: v.value.clone() for v in model.trainable_variables}
        out = step(f, l)
        torch.cuda.synchronize()
        assert torch.isfinite(out['total_loss']).all()
        k = model.supervised_head.linear_layer.kernel
        gk = k.grad.double().cpu().numpy()
        w0 = before[k.name].double().cpu().numpy()
        if optimizer == 'adam':
            want, _, _ = oo.adam_apply(w0, gk, 0 * w0, 0 * w0, 0.01, 1, l2=1e-3)
        else:
            want, _ = oo.sgd_apply(w0, gk, 0 * w0, 0.01, 0.9, True, l2=1e-3)
        assert np.abs(k.value.double().cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max() + 1e-8
        moved = [v.name for v in model.trainable_variables if not torch.equal(before[v.name], v.value)]
        # what may stand still: bn1's gamma / beta of the 8 residual blocks -- their gradient passes through the zero-initialised tail
        # gamma (exactly zero at step 0) and BatchNorm variables take no weight-decay term
        assert len(moved) >= len(before) - 16, sorted(set(before) - set(moved))
    finally:
        FLAGS.reset()
        RT.reset()
