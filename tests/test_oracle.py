"""CPU tests: the oracle against closed forms, golden vectors, the reference's structural known
answers, and float64 self-consistency (analytic gradient == autograd; sharded == global batch)."""
import os

import numpy as np
import pytest
import torch

from oracle import lars as olars
from oracle import ntxent as ont
from oracle.model_torch import Config, init_model, torch_contrastive_loss, train_step

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


# ---- closed-form known answers -----------------------------------------------
@pytest.mark.parametrize('n', [4, 16, 64])
def test_ntxent_identical_rows(n):
    loss, _, _ = ont.add_contrastive_loss(np.ones((2 * n, 32)), True, 0.1)
    assert abs(loss - 2 * np.log(2 * n - 1)) < 1e-12


@pytest.mark.parametrize('T', [0.1, 0.5, 1.0])
def test_ntxent_orthogonal_onehot(T):
    n = 16
    e = np.eye(n, 32)
    loss, _, _ = ont.add_contrastive_loss(np.concatenate([e, e]), True, T)
    closed = 2 * (np.log(np.exp(1 / T) + (2 * n - 2)) - 1 / T)
    assert abs(loss - closed) < 1e-12


def test_ntxent_tiny_by_hand():
    # hidden_norm=False, T=1, n=1: logits_ab=[h1.h2], logits_aa masked -> loss = 2*(lse([s, -1e9]) - s) = 0
    h = np.array([[1.0, 2.0], [3.0, -1.0]])
    loss, lab, labels = ont.add_contrastive_loss(h, False, 1.0)
    assert abs(loss) < 1e-9 and lab.shape == (1, 1) and labels.shape == (1, 2)
    # n=2 by hand
    h = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 2.0]])
    h1, h2 = h[:2], h[2:]
    def xent(row_logits, pos):
        return np.log(np.sum(np.exp(row_logits))) - row_logits[pos]
    la = [xent(np.array([h1[i] @ h2[0], h1[i] @ h2[1], h1[i] @ h1[1 - i]]), i) for i in range(2)]
    lb = [xent(np.array([h2[i] @ h1[0], h2[i] @ h1[1], h2[i] @ h2[1 - i]]), i) for i in range(2)]
    loss, _, _ = ont.add_contrastive_loss(h, False, 1.0)
    assert abs(loss - np.mean(np.array(la) + np.array(lb))) < 1e-9


def test_ntxent_permutation_invariance():
    g = np.random.default_rng(0)
    n = 8
    h = g.standard_normal((2 * n, 16))
    p = g.permutation(n)
    hp = np.concatenate([h[:n][p], h[n:][p]])
    assert abs(ont.add_contrastive_loss(h, True, 0.1)[0] - ont.add_contrastive_loss(hp, True, 0.1)[0]) < 1e-12


# ---- float64 self-consistency ---------------------------------------------------------------------
def test_analytic_grad_matches_autograd():
    g = np.random.default_rng(1)
    h = g.standard_normal((24, 32))
    losses, grads = ont.contrastive_loss_and_grad([h], True, 0.2)
    ht = torch.tensor(h, requires_grad=True)
    l, _, _, _ = torch_contrastive_loss(ht, True, 0.2)
    l.backward()
    assert abs(losses[0] - float(l.detach())) < 1e-12
    assert np.abs(grads[0] - ht.grad.numpy()).max() < 1e-14


@pytest.mark.parametrize('R', [1, 2, 4, 8])
def test_sharded_equals_global_batch(R):
    """R replicas with differentiable concat + loss/R + gradient SUM == 1 replica on the global
    batch (tf2/objective.py:60-69, tf2/run.py:617-622)."""
    g = np.random.default_rng(2)
    n = 4
    hs = [g.standard_normal((2 * n, 16)) for _ in range(R)]
    losses, grads = ont.contrastive_loss_and_grad(hs, True, 0.1)
    glob = np.concatenate([x[:n] for x in hs] + [x[n:] for x in hs])
    lg, gg = ont.contrastive_loss_and_grad([glob], True, 0.1)
    assert abs(np.mean(losses) - lg[0]) < 1e-12
    N = R * n
    for r in range(R):
        ref = np.concatenate([gg[0][r * n:(r + 1) * n], gg[0][N + r * n:N + (r + 1) * n]])
        assert np.abs(grads[r] - ref).max() < 1e-14
        lr, _, _ = ont.add_contrastive_loss(hs[r], True, 0.1, all_hiddens=hs if R > 1 else None, replica_id=r)
        assert abs(lr - losses[r]) < 1e-12


def test_cross_replica_concat_is_concat():
    ts = [np.full((2, 3), float(i)) for i in range(4)]
    assert np.array_equal(ont.tpu_cross_replica_concat(ts), np.concatenate(ts))
    assert ont.tpu_cross_replica_concat(ts[:1]) is ts[0]


# ---- golden vectors ----------------------------------------------------------------------------------
def test_ntxent_golden():
    z = np.load(os.path.join(GOLD, 'ntxent_golden.npz'))
    for name in 'abcd':
        n, R, D, T, norm = z[name + '_meta']
        hs = list(z[name + '_hidden'])
        losses, grads = ont.contrastive_loss_and_grad(hs, bool(norm), float(T))
        assert np.abs(np.array(losses) - z[name + '_loss']).max() < 1e-12
        assert np.abs(np.stack(grads) - z[name + '_grad']).max() < 1e-13


def test_lars_golden_and_rules():
    z = np.load(os.path.join(GOLD, 'lars_golden.npz'))
    names = open(os.path.join(GOLD, 'lars_names.txt')).read().split()
    ex = ['batch_normalization', 'bias', 'head_supervised']
    for i, name in enumerate(names):
        w, g, v = z['t%d_in' % i]
        for classic in (True, False):
            for nest in (False, True):
                nw, nv = olars.lars_apply(name, w, g, v, 0.3, momentum=0.9, use_nesterov=nest, weight_decay=1e-4,
                                          exclude_from_weight_decay=ex, classic_momentum=classic)
                key = 't%d_c%d_n%d' % (i, classic, nest)
                assert np.abs(nw - z[key + '_w']).max() < 1e-15
                assert np.abs(nv - z[key + '_v']).max() < 1e-15
    # exclusion rules (tf2/lars_optimizer.py:139-157, tf2/model.py:36-42); re.search = substring
    assert not olars.use_weight_decay('x/sync_batch_normalization_3/gamma:0', 1e-4, ex)
    assert not olars.use_weight_decay('head_supervised/linear_layer/dense/kernel:0', 1e-4, ex)
    assert olars.use_weight_decay('resnet/conv2d_1/kernel:0', 1e-4, ex)
    assert not olars.use_weight_decay('resnet/conv2d_1/kernel:0', 0.0, ex)
    assert not olars.do_layer_adaptation('dense/bias:0', ex)
    # ||w|| = 0 or ||g|| = 0 -> trust ratio 1 (plain momentum step)
    w = np.zeros(5); g = np.ones(5); v = np.zeros(5)
    nw, nv = olars.lars_apply('k/kernel:0', w, g, v, 0.5, weight_decay=0.0, exclude_from_weight_decay=ex)
    assert np.allclose(nv, 0.5 * g) and np.allclose(nw, -0.5 * g)
    nw, nv = olars.lars_apply('k/kernel:0', np.ones(5), np.zeros(5), v, 0.5, weight_decay=0.0,
                              exclude_from_weight_decay=ex)
    assert np.allclose(nw, 1.0)


def test_lr_schedule():
    kw = dict(warmup_epochs=10, train_batch_size=4096, learning_rate_scaling='linear', train_epochs=100)
    f = lambda s: olars.warmup_and_cosine_decay(s, 0.3, 1281167, **kw)
    warm = int(round(10 * 1281167 // 4096))
    assert warm == 3127 and f(0) == 0.0
    assert abs(f(warm) - 0.3 * 4096 / 256) < 1e-12
    total = 1281167 * 100 // 4096 + 1
    assert abs(f(total)) < 1e-9 and f(warm // 2) == pytest.approx(0.5 * (warm // 2 * 2) / warm * 4.8, rel=1e-3)
    assert olars.warmup_and_cosine_decay(10 ** 6, 0.075, 1281167, train_batch_size=4096,
                                         learning_rate_scaling='sqrt') == 0.0
    assert olars.get_train_steps(1000, 7, 100, 10) == 7


# ---- structural known answers published by the reference ------------------------------------------------
@pytest.mark.parametrize('depth,width,sk,millions', [
    (50, 1, 0.0, 23.56),      # README.md:21  "24", colabs/load_and_inference.ipynb:406 "23.56M"
    (50, 1, 0.0625, 35.28),   # README.md:22  "35"
    (50, 2, 0.0, 94.01),      # README.md:23  "94"
])
def test_param_counts_match_readme(depth, width, sk, millions):
    cfg = Config(resnet_depth=depth, width_multiplier=width, sk_ratio=sk, image_size=64,
                 lineareval_while_pretraining=False, proj_head_mode='none')
    p, s = init_model(cfg)
    n = sum(v.numel() for v in p.values()) + sum(v.numel() for v in s.values())
    assert round(n / 1e6, 2) == millions


def test_endpoint_shapes_r50():
    """tf2/colabs/finetuning.ipynb:909 (scaled to 64 px: /3.5)."""
    from oracle.model_torch import Builder
    cfg = Config(resnet_depth=50, image_size=64, lineareval_while_pretraining=False)
    p, s = init_model(cfg)
    b = Builder(cfg, params=p, state=s)
    with torch.no_grad():
        b.model(torch.rand(1, 64, 64, 6))
    shp = {k: tuple(v.shape) for k, v in b.endpoints.items()}
    assert shp['initial_conv'] == (2, 64, 32, 32) and shp['initial_max_pool'] == (2, 64, 16, 16)
    assert shp['block_group1'] == (2, 256, 16, 16) and shp['block_group4'] == (2, 2048, 2, 2)
    assert shp['final_avg_pool'] == (2, 2048)
    k = [n for n in p if n.endswith('kernel:0')]
    assert tuple(p[k[0]].shape) == (7, 7, 3, 64)                       # HWIO
    assert tuple(p['model/projection_head/nl_2/dense_2/kernel:0'].shape) == (2048, 128)   # [in, out]


def test_train_step_runs_and_updates_bn():
    from collections import OrderedDict
    cfg = Config(resnet_depth=18, image_size=32, num_classes=10)
    p, s = init_model(cfg, randomize_bn=True)
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4, 32, 32, 6, generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (4,), generator=g), 10).float()
    np_, ns, nm, info = train_step(cfg, p, s, m, x, y, 0.1)
    assert torch.isfinite(info['total_loss'])
    # sup-head gradient never reaches the encoder (stop_gradient, tf2/model.py:276-277):
    k = 'model/head_supervised/linear_layer/dense_3/kernel:0'
    assert info['grads'][k].abs().max() > 0
    mm = [n for n in s if 'moving_mean' in n][0]
    assert not torch.equal(ns[mm], s[mm])


def test_blur_oracle_is_separable_gaussian_and_selector_semantics():
    """gaussian_blur == 2-D convolution with the outer product of the 1-D filter (zero padding);
    unselected images pass through (clipped)."""
    from oracle import blur as oblur
    g = np.random.default_rng(0)
    x = g.random((2, 9, 9, 3))
    y = oblur.gaussian_blur(x, 5, 0.8)
    r = 2
    t = np.arange(-r, r + 1)
    f = np.exp(-t ** 2 / (2 * 0.8 ** 2)); f /= f.sum()
    k2 = np.outer(f, f)
    pad = np.zeros((2, 13, 13, 3)); pad[:, 2:11, 2:11] = x
    ref = sum(k2[i, j] * pad[:, i:i + 9, j:j + 9] for i in range(5) for j in range(5))
    assert np.abs(y - ref).max() < 1e-12
    out = oblur.batch_random_blur([x * 1.5], 50, [1.0], [np.array([0.0, 1.0])])[0]
    assert np.array_equal(out[0], np.clip(x[0] * 1.5, 0, 1)) and not np.array_equal(out[1], np.clip(x[1] * 1.5, 0, 1))


# ---- hand-derived known answers (tests/golden/HAND_DERIVED.md; NOT oracle output) ----------------
def _hand():
    import json
    return json.load(open(os.path.join(GOLD, 'hand_derived.json')))


def test_hand_derived_lars_cases():
    """tf2/lars_optimizer.py:83-137 worked by hand: weight decay, trust ratio, classic / Nesterov / 'popular' momentum,
    the excluded-name filters and both zero-norm branches."""
    h = _hand()['lars']
    for c in h['cases']:
        nw, nv = olars.lars_apply(c['name'], np.array(c['w']), np.array(c['g']), np.array(c['v']), h['lr'],
                                  momentum=h['momentum'], use_nesterov=c['nesterov'],
                                  weight_decay=c.get('weight_decay', h['weight_decay']),
                                  exclude_from_weight_decay=h['exclude_from_weight_decay'],
                                  classic_momentum=c['classic'], eeta=h['eeta'])
        assert np.allclose(nw, c['w_new'], rtol=0, atol=1e-12), (c, nw)
        assert np.allclose(nv, c['v_new'], rtol=0, atol=1e-12), (c, nv)


def bn_hand_expected():
    """Evaluates the closed-form expressions of tests/golden/hand_derived.json (float64)."""
    from math import sqrt  # noqa: F401  (used by eval below)
    b = _hand()['batch_norm']
    env = {'sqrt': sqrt}
    env.update({k: eval(v, {'sqrt': sqrt}) for k, v in b['r_expr'].items()})
    y = np.array([[eval(e, dict(env)) for e in row] for row in b['y_relu_expr']], dtype=np.float64)
    x = np.array(b['x'])
    var = np.array(b['biased_var'])
    xh = (x - np.array(b['mean'])) / np.sqrt(var + b['eps'])
    bwd = dict(dy=xh, dbeta=np.zeros(2), dgamma=4 * var / (var + b['eps']),
               dx=np.array(b['gamma']) / np.sqrt(var + b['eps']) * xh * b['eps'] / (var + b['eps']))
    return b, y, bwd


def test_hand_derived_batch_norm():
    """tf2/resnet.py:31-78: biased variance, moving-average update, BN+ReLU output, and the BN backward."""
    from oracle.model_torch import Builder
    b, y_ref, bwd = bn_hand_expected()
    x = torch.tensor(b['x'], dtype=torch.float64).t().reshape(1, 2, 4, 1).permute(2, 1, 0, 3).contiguous()   # [N=4, C=2, 1, 1]
    bl = Builder(Config(), dtype=torch.float64)
    bl.batch_norm_relu(x, relu=True)                       # creates gamma=1, beta=0; overwrite, then run
    names = list(bl.params.keys())
    bl.params[names[0]] = torch.tensor(b['gamma'], dtype=torch.float64)
    bl.params[names[1]] = torch.tensor(b['beta'], dtype=torch.float64)
    bl.init = False
    bl.namer = type(bl.namer)()
    y = bl.batch_norm_relu(x, relu=True)[:, :, 0, 0].numpy()
    assert np.abs(y - y_ref).max() < 1e-12
    st = {k.rsplit('/', 1)[1]: v.numpy() for k, v in bl.new_state.items()}
    assert np.allclose(st['moving_mean:0'], b['moving_mean'], atol=1e-15)
    assert np.allclose(st['moving_variance:0'], b['moving_variance'], atol=1e-15)
    # backward of the BN alone for dy = x^
    xr = x.clone().requires_grad_(True)
    g = bl.params[names[0]].clone().requires_grad_(True)
    be = bl.params[names[1]].clone().requires_grad_(True)
    bl.params[names[0]], bl.params[names[1]] = g, be
    bl.namer = type(bl.namer)()
    out = bl.batch_norm_relu(xr, relu=False)
    out.backward(torch.tensor(bwd['dy']).reshape(4, 2, 1, 1))
    assert np.abs(xr.grad[:, :, 0, 0].numpy() - bwd['dx']).max() < 1e-12
    assert np.abs(g.grad.numpy() - bwd['dgamma']).max() < 1e-12 and np.abs(be.grad.numpy() - bwd['dbeta']).max() < 1e-12


def test_oracle_matches_tensorflow_reference():
    """Runs /root/reference/tf2 itself against the oracle when TensorFlow is importable (it is not in this image)."""
    from oracle import check_against_tf as chk
    if not chk.available():
        pytest.skip('a real tensorflow is not importable here: the oracle is pinned to the reference SOURCE on a numpy stand-in instead (tests/test_reference_pin.py, DESIGN.md section 5)')
    ok, res = chk.run_all()
    assert ok, res


def test_hand_derived_ntxent_swapped_views():
    """tests/golden/HAND_DERIVED.md, last section: loss 2 log(2n-2+e), accuracy 0, entropy log(n-1+e) - e/(n-1+e)."""
    import math
    from oracle import ntxent as ont
    n = 64
    e = np.eye(n)
    h = np.concatenate([e, e[np.arange(n) ^ 1]], 0)
    loss, logits_ab, labels = ont.add_contrastive_loss(h, True, 1.0)
    acc, ent = ont.contrastive_metrics(logits_ab, labels)
    assert abs(float(loss) - 2 * math.log(2 * n - 2 + math.e)) < 1e-9
    assert acc == 0.0
    assert abs(ent - (math.log(n - 1 + math.e) - math.e / (n - 1 + math.e))) < 1e-6
