"""The oracle against the REFERENCE'S OWN SOURCE (tests/golden/reference_pin.npz).

The fixtures were produced in the build container by importing /root/reference/tf2/{objective,lars_optimizer,metrics,resnet,
data_util,model}.py unmodified on top of oracle/tfshim.py (a float64 numpy stand-in for the TensorFlow / Keras calls they make) --
tests/golden/make_reference_golden.py.  Here: (1) every fixture is recomputed with oracle/*.py and must agree; (2) wherever
the reference checkout exists, the fixtures are regenerated from it in a subprocess and must be reproduced exactly."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = os.path.join(HERE, 'golden', 'make_reference_golden.py')


def _script():
    spec = importlib.util.spec_from_file_location('make_reference_golden', SCRIPT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_the_reference_source_fixtures():
    m = _script()
    ref = dict(np.load(m.OUT_NPZ))
    orc = m.oracle_cases()
    assert set(ref) == set(orc), (sorted(set(ref) - set(orc))[:5], sorted(set(orc) - set(ref))[:5])
    assert len(ref) >= 241
    bad = [(k, m.compare(ref[k], orc[k], k)) for k in sorted(ref) if m.compare(ref[k], orc[k], k) > m.tolerance(k)]
    assert not bad, bad[:10]
    # what the table covers (a fixture file that silently lost a family would pass the loop above)
    fam = {k.split('_')[0].rstrip('0123456789') for k in ref}
    assert {'ntx', 'ntxr', 'sup', 'lars', 'sched', 'blur', 'conv', 'bn', 'r', 'aug', 'augeval', 'step'} <= fam, fam
    for tag in ('r18_cifar', 'r50', 'r50_sk', 'r34_w2', 'r18_localbn'):
        assert len(ref[tag + '_names']) == {'r18_cifar': 121, 'r50': 281, 'r50_sk': 387, 'r34_w2': 196, 'r18_localbn': 119}[tag]
        assert list(ref[tag + '_names']) == list(orc[tag + '_names'])       # variable names as the reference's layer construction yields them


def test_variable_names_follow_the_lars_name_filters():
    """the name filters of tf2/model.py:39-41 ('batch_normalization', 'bias', 'head_supervised') meet the names the reference's own
    layer construction produced: BatchNorm variables carry 'sync_batch_normalization' (global_bn), the linear-eval head 'head_supervised'"""
    ref = dict(np.load(_script().OUT_NPZ))
    names = list(ref['r50_names'])
    bn = [n for n in names if n.endswith(('gamma:0', 'beta:0'))]
    assert bn and all('batch_normalization' in n for n in bn)
    assert [n for n in names if 'head_supervised' in n] == ['head_supervised/linear_layer/dense_3/bias:0', 'head_supervised/linear_layer/dense_3/kernel:0']
    assert sum(n.endswith('kernel:0') for n in names) == 53 + 3 + 1          # 53 convolutions, 3 projection-head layers, the linear-eval head


@pytest.mark.skipif(not os.path.isdir(os.path.join(os.environ.get('SIMCLR_REFERENCE', '/root/reference'), 'tf2')),
                    reason='reference checkout not present (GPU box): the committed fixtures are used as they are')
def test_fixtures_are_reproduced_by_the_reference_source():
    # a subprocess: the stand-in registers itself as `tensorflow`, which must not leak into this interpreter
    r = subprocess.run([sys.executable, SCRIPT, '--check'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'tensorflow' not in sys.modules or not getattr(sys.modules['tensorflow'], '_SIMCLR_SHIM', False)
