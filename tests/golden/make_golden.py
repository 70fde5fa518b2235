"""Generates the committed golden vectors under tests/golden/.

The reference (google-research/simclr) ships NO tests or fixtures and TensorFlow is not
installable here (no network), so these vectors cannot come from the reference itself: they are
float64 outputs of the oracle (oracle/ntxent.py, oracle/lars.py -- line-by-line restatements of
tf2/objective.py and tf2/lars_optimizer.py) on seeded inputs, plus closed-form known answers.
They pin the oracle against regressions and give the HIP path fixed input/output pairs.
"PARITY UNPINNED" (see oracle/__init__.py) still applies.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lars as olars  # noqa: E402
from oracle import ntxent as ont  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ntxent_cases():
    out = {}
    rng = np.random.default_rng(3)
    for name, n, R, D, T, norm in [('a', 8, 1, 64, 0.1, True), ('b', 16, 2, 128, 0.1, True),
                                   ('c', 4, 4, 64, 0.5, True), ('d', 8, 1, 64, 1.0, False)]:
        hs = [rng.standard_normal((2 * n, D)).astype(np.float32) for _ in range(R)]
        losses, grads = ont.contrastive_loss_and_grad(hs, norm, T)
        accs, ents = [], []
        for r in range(R):
            _, lab, labels = ont.add_contrastive_loss(hs[r], norm, T, all_hiddens=hs if R > 1 else None, replica_id=r)
            a, e = ont.contrastive_metrics(lab, labels)
            accs.append(a); ents.append(e)
        out['%s_hidden' % name] = np.stack(hs)
        out['%s_meta' % name] = np.array([n, R, D, T, float(norm)])
        out['%s_loss' % name] = np.array(losses)
        out['%s_grad' % name] = np.stack(grads)
        out['%s_acc' % name] = np.array(accs)
        out['%s_entropy' % name] = np.array(ents)
    np.savez_compressed(os.path.join(HERE, 'ntxent_golden.npz'), **out)


def lars_cases():
    rng = np.random.default_rng(7)
    out = {}
    ex = ['batch_normalization', 'bias', 'head_supervised']
    names = ['conv2d/kernel:0', 'sync_batch_normalization/gamma:0', 'dense/bias:0',
             'head_supervised/linear_layer/dense_3/kernel:0', 'zero/kernel:0']
    for i, name in enumerate(names):
        w = (rng.standard_normal(257) * 0.05).astype(np.float32)
        if name.startswith('zero'):
            w[:] = 0
        g = (rng.standard_normal(257) * 1e-3).astype(np.float32)
        v = (rng.standard_normal(257) * 1e-3).astype(np.float32)
        for classic in (True, False):
            for nest in (False, True):
                nw, nv = olars.lars_apply(name, w, g, v, 0.3, momentum=0.9, use_nesterov=nest, weight_decay=1e-4,
                                          exclude_from_weight_decay=ex, classic_momentum=classic)
                key = 't%d_c%d_n%d' % (i, classic, nest)
                out[key + '_w'] = nw; out[key + '_v'] = nv
        out['t%d_in' % i] = np.stack([w, g, v])
    np.savez_compressed(os.path.join(HERE, 'lars_golden.npz'), **out)
    with open(os.path.join(HERE, 'lars_names.txt'), 'w') as f:
        f.write('\n'.join(names) + '\n')


if __name__ == '__main__':
    ntxent_cases()
    lars_cases()
    print('wrote', sorted(os.listdir(HERE)))
