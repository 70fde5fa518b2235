"""CPU tests of the C-ABI boundary and the host-side logic (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

from simclr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    sig = _lib.parse_header()
    assert len(sig) >= 30
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in sig:
        assert hasattr(dll, name), 'libsimclr_hip.so does not export %s' % name
    L = _lib.lib()
    assert L.abi_version() == _lib.ABI_VERSION == 8 and L.lars_chunk_elems() == 8192
    assert L.ntxent_workspace_bytes(512, 512, 128) > 0
    assert L.conv2d_wgrad_workspace_bytes(8, 56, 56, 64, 64, 3, 3, _lib.DT_BF16) > 0


def test_no_undeclared_exports():
    """Every exported simclr_* symbol is declared in the header (the header is the whole boundary)."""
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r'\bT (simclr_\w+)', out))
    declared = set(_lib.parse_header())
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_bad_arguments_return_error_codes():
    L = _lib.lib()
    with pytest.raises(_lib.SimclrHipError, match='D must be 64/128/256'):
        L.ntxent_fwd(None, None, 4, 4, 100, 0, ctypes.c_float(0.1), None, None, None, None)
    with pytest.raises(_lib.SimclrHipError, match='multiple of'):
        L.conv2d_fwd(None, None, None, None, 0, 1, 8, 8, 3, 8, 8, 64, 3, 3, 1, 1, _lib.DT_BF16, None)
    # round 4 entry points: pivoted BatchNorm statistics are an fp32 feature; the slot conversion needs a row count
    with pytest.raises(_lib.SimclrHipError, match='fp32 only'):
        L.conv2d_fwd_pivoted(None, None, None, None, 4, None, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, _lib.DT_BF16, None)
    with pytest.raises(_lib.SimclrHipError, match='null argument'):
        L.conv2d_fwd_pivoted(None, None, None, None, 4, None, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, _lib.DT_F32, None)
    with pytest.raises(_lib.SimclrHipError, match='bad shape'):
        L.bn_reduce_slots_pivoted(None, 4, 64, None, ctypes.c_double(0.0), None, None)
    assert L.conv2d_last_presplit() == 0
    with pytest.raises(_lib.SimclrHipError, match='empty tensor list'):
        L.lars_multi_tensor(None, 0, None, 0, None, ctypes.c_float(0.1), ctypes.c_float(0.9),
                            ctypes.c_float(0.0), ctypes.c_float(0.001), 1, 0, None, None)


def test_product_path_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'simclr_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_flags_match_reference_defaults():
    from simclr_amd.flags import FLAGS
    FLAGS.reset()
    d = FLAGS.flag_values_dict()
    ref = dict(learning_rate=0.3, learning_rate_scaling='linear', warmup_epochs=10, weight_decay=1e-6,
               batch_norm_decay=0.9, train_batch_size=512, train_epochs=100, temperature=0.1, hidden_norm=True,
               proj_head_mode='nonlinear', proj_out_dim=128, num_proj_layers=3, global_bn=True, width_multiplier=1,
               resnet_depth=50, sk_ratio=0.0, image_size=224, use_blur=True, optimizer='lars', momentum=0.9,
               lineareval_while_pretraining=True, train_mode='pretrain')
    for k, v in ref.items():
        assert d[k] == v, k
    FLAGS.parse(['--train_batch_size=4096', '--nouse_blur', '--hidden_norm=False', '--resnet_depth', '18'])
    assert FLAGS.train_batch_size == 4096 and FLAGS.use_blur is False and FLAGS.hidden_norm is False
    assert FLAGS.resnet_depth == 18
    FLAGS.reset()


def test_schedule_and_lars_rules_match_oracle():
    from oracle import lars as olars
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    FLAGS.reset()
    FLAGS.update(train_batch_size=4096, learning_rate_scaling='sqrt')
    sched = model_lib.WarmUpAndCosineDecay(0.075, 1281167)
    for s in (0, 1, 500, 3126, 3127, 3128, 10000, 31279, 40000):
        assert sched(s) == pytest.approx(olars.warmup_and_cosine_decay(
            s, 0.075, 1281167, train_batch_size=4096, learning_rate_scaling='sqrt'), abs=1e-12)
    assert model_lib.get_train_steps(1281167) == olars.get_train_steps(1281167, 0, 100, 4096)
    opt = model_lib.build_optimizer(0.1)
    ex = ['batch_normalization', 'bias', 'head_supervised']
    for name in ['a/conv2d/kernel:0', 'a/sync_batch_normalization_2/beta:0', 'head_supervised/x/dense/kernel:0',
                 'proj/dense/bias:0']:
        assert opt._use_weight_decay(name) == olars.use_weight_decay(name, FLAGS.weight_decay, ex)
        assert opt._do_layer_adaptation(name) == olars.do_layer_adaptation(name, ex)
    FLAGS.reset()


def test_model_constructs_on_cpu_and_names_layers():
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    FLAGS.reset(); FLAGS.update(use_blur=False)
    RT.reset()
    m = model_lib.Model(1000)
    assert RT.counters['conv2d'] == 53 and RT.counters['sync_batch_normalization'] == 56
    assert RT.counters['bottleneck_block'] == 16 and RT.counters['dense'] == 4
    FLAGS.update(sk_ratio=0.0625); RT.reset()
    model_lib.Model(1000)        # SK + ResNet-D: 16 SK units, 2 extra stem convs, 32 bare 1x1 convs
    assert RT.counters['sk__conv2d'] == 16 and RT.counters['conv2d_fixed_padding'] == 55
    assert RT.counters['conv2d'] == 87
    FLAGS.update(sk_ratio=0.0, se_ratio=0.25); RT.reset()
    with pytest.raises(NotImplementedError):
        model_lib.Model(1000)
    FLAGS.reset(); RT.reset()


def test_header_is_plain_c():
    """include/simclr_hip.h must be consumable by a C compiler (it is what cgo / JNI / ctypes-style bindings read)."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'hdr.c')
        with open(src, 'w') as f:
            f.write('#include "simclr_hip.h"\nint main(void) { return simclr_abi_version() == 0; }\n')
        r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(root, 'include'), '-fsyntax-only', src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_metric_mean_with_attached_bank():
    """metrics.Mean keeps tf.keras.metrics.Mean semantics when its running sum lives in the step's shared device tensor
    (run.make_single_step: one simclr_accumulate_scalars launch per step updates every attached metric)."""
    import torch
    from simclr_amd.metrics import Mean
    bank = torch.zeros(4)
    m, other = Mean('a'), Mean('b')
    m.attach(bank, 2)
    other.attach(bank, 0)
    assert m.result() == 0.0
    for v in (1.0, 2.0, 6.0):          # what the kernel does: bank[i] += value; the host only counts
        bank[2] += v
        m.bump()
    assert m.result() == 3.0 and other.result() == 0.0
    m.update_state(torch.tensor([3.0]))                       # the stand-alone path still works next to the bank
    assert m.result() == 3.0
    m.reset_states()
    assert m.result() == 0.0 and float(bank[2]) == 0.0 and m._count == 0


def test_class_id_cache_follows_the_label_tensor():
    import torch
    from simclr_amd.objective import _class_ids
    lab = torch.nn.functional.one_hot(torch.tensor([2, 0, 1]), 3).float()
    a = _class_ids(lab)
    assert a.dtype == torch.int32 and a.tolist() == [2, 0, 1]
    assert _class_ids(lab) is a                                # same storage, same version: converted once
    lab[0] = torch.tensor([1.0, 0.0, 0.0])                     # in-place change bumps the version -> recomputed
    assert _class_ids(lab).tolist() == [0, 0, 1]
    assert _class_ids(torch.tensor([1, 2])).tolist() == [1, 2]


def test_instantiation_table_matches_the_committed_one():
    """which kernel instantiation every (ResNet-50 layer class, launch kind, storage mode) selects is ONE table,
    produced without a GPU (SIMCLR_DRY_RUN=1: launch_igemm_one takes its decisions and records them instead of launching) and
    committed under profiles/ -- a change of a selection rule shows up here as a diff of that file."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a subprocess: the library caches a few environment switches at first use
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'instantiation_table.py')], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if not k.startswith('SIMCLR_')})
    assert r.returncode == 0, r.stderr[-2000:]
    want = open(os.path.join(root, 'profiles', 'r06_instantiations.txt')).read()
    assert r.stdout == want, 'the (layer class -> instantiation) table changed: regenerate profiles/r06_instantiations.txt with tools/instantiation_table.py and review the diff'
    rows = [l for l in want.splitlines() if l and not l.startswith('#')]
    assert len(rows) > 200 and any('conv_igemm_wide' in l for l in rows) and any('elt=4' in l and ', 6>' in l for l in rows)


def test_product_defaults_are_the_tolerance_meeting_mode():
    """a user who switches to this package gets the mode whose outputs meet north_star's tolerances -- fp32 storage,
    three fp16-piece terms forward / three bf16-piece terms backward -- unless they opt into the bf16 speed mode.  (The test session itself
    overrides the f32_matmul DEFAULT to 'exact' through SIMCLR_DEFAULT_F32_MATMUL: checked in a clean subprocess.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != 'SIMCLR_DEFAULT_F32_MATMUL'}
    out = subprocess.run([sys.executable, '-c', 'from simclr_amd.flags import FLAGS; print(FLAGS.compute_dtype, FLAGS.f32_matmul, FLAGS.ntxent_matmul)'],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['f32', 'f16x3_3', 'exact']
