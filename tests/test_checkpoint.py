"""Checkpoint / resume host logic (tf2/run.py:308-337) on CPU with a stand-in model; eval metrics."""
import json
import os

import pytest
import torch

from simclr_amd import metrics
from simclr_amd.checkpoint import Checkpoint, CheckpointManager, try_restore_from_checkpoint
from simclr_amd.lars_optimizer import LARSOptimizer, Variable
from simclr_amd.resnet import RT, Layer


class _Head(Layer):
    def __init__(self):
        self.kernel = Variable('head_supervised/linear_layer/dense/kernel:0', torch.randn(4, 3))
        self.bias = Variable('head_supervised/linear_layer/dense/bias:0', torch.randn(3))


class _Toy(Layer):
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.w = Variable('conv2d/kernel:0', torch.randn(3, 3, 2, 4, generator=g))
        self.mm = Variable('batch_normalization/moving_mean:0', torch.randn(4, generator=g), trainable=False)
        self.supervised_head = _Head()


def _opt(model, iterations):
    opt = LARSOptimizer(0.1)
    opt._create_slots(model.variables)
    for v in model.variables:
        opt._slots[id(v)].copy_(torch.randn(v.shape))
    opt.iterations = iterations
    return opt


def test_save_restore_roundtrip_and_retention(tmp_path):
    d = str(tmp_path / 'run')
    m = _Toy(0)
    opt = _opt(m, 7)
    mgr = CheckpointManager(Checkpoint(model=m, optimizer=opt), d, max_to_keep=2)
    assert mgr.latest_checkpoint is None
    paths = []
    for step in (7, 14, 21):
        opt.iterations = step
        paths.append(mgr.save(step))
    assert [os.path.basename(p) for p in mgr.checkpoints] == ['ckpt-14.pt', 'ckpt-21.pt']     # keep_checkpoint_max
    assert not os.path.exists(paths[0]) and os.path.exists(paths[2])
    idx = json.load(open(os.path.join(d, 'checkpoint.json')))
    assert idx['model_checkpoint_path'] == 'ckpt-21.pt'

    # a fresh process: new objects, different values -> restored bit-exactly, weights_version bumped
    m2 = _Toy(1)
    opt2 = _opt(m2, 0)
    ver = RT.weights_version
    mgr2, status = try_restore_from_checkpoint(m2, opt2, d)
    assert mgr2.latest_checkpoint.endswith('ckpt-21.pt')
    status.assert_consumed()
    assert RT.weights_version == ver + 1
    assert opt2.iterations == 21
    for a, b in zip(m.variables, m2.variables):
        assert a.name == b.name and torch.equal(a.value, b.value)
        assert torch.equal(opt._slots[id(a)], opt2._slots[id(b)])


def test_restore_from_given_checkpoint_is_weights_only_and_can_zero_the_head(tmp_path):
    d = str(tmp_path / 'pretrained')
    m = _Toy(0)
    opt = _opt(m, 99)
    path = CheckpointManager(Checkpoint(model=m, optimizer=opt), d).save()
    assert path.endswith('ckpt-99.pt')
    m2 = _Toy(1)
    opt2 = _opt(m2, 5)
    slot_before = opt2._slots[id(m2.w)].clone()
    mgr, status = try_restore_from_checkpoint(m2, opt2, str(tmp_path / 'empty_model_dir'), checkpoint=path,
                                              zero_init_logits_layer=True)
    assert mgr.latest_checkpoint is None and status is not None
    assert torch.equal(m2.w.value, m.w.value) and torch.equal(m2.mm.value, m.mm.value)
    assert opt2.iterations == 5 and torch.equal(opt2._slots[id(m2.w)], slot_before)     # run.py:320-327: weights only
    assert float(m2.supervised_head.kernel.value.abs().sum()) == 0.0                   # run.py:329-335
    assert float(m2.supervised_head.bias.value.abs().sum()) == 0.0


def test_partial_and_mismatched_checkpoints(tmp_path):
    m = _Toy(0)
    path = str(tmp_path / 'c.pt')
    Checkpoint(model=m).write(path)
    m2 = _Toy(1)
    m2.extra = Variable('new_layer/kernel:0', torch.zeros(2))
    st = Checkpoint(model=m2).restore(path)
    assert st.missing_in_checkpoint == ['new_layer/kernel:0'] and not st.unused_in_checkpoint
    st.expect_partial()
    with pytest.raises(AssertionError):
        st.assert_consumed()
    m3 = _Toy(1)
    m3.w = Variable('conv2d/kernel:0', torch.zeros(1, 1, 2, 4))
    with pytest.raises(ValueError):
        Checkpoint(model=m3).restore(path).expect_partial()
    with pytest.raises(ValueError):
        m.dup = Variable('conv2d/kernel:0', torch.zeros(1))
        Checkpoint(model=m).state_dict()


def test_eval_metrics_match_keras_definitions():
    logits = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0, 0.0, 0.0],
                           [0.9, 0.1, 0.8, 0.7, 0.6, 0.5, 0.4],
                           [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                           [0.5, 0.4, 0.3, 0.2, 0.1, 0.6, 0.0]])
    labels = torch.nn.functional.one_hot(torch.tensor([1, 6, 6, 3]), 7).float()
    top1 = metrics.Accuracy('eval/label_top_1_accuracy')
    top5 = metrics.TopKCategoricalAccuracy(5, 'eval/label_top_5_accuracy')
    metrics.update_finetune_metrics_eval(top1, top5, logits, labels)
    assert top1.result() == pytest.approx(2 / 4)          # rows 0 and 2
    # row 1: target 6 has 5 larger entries -> out of the top 5; row 3: target 3 has 4 larger -> in
    assert top5.result() == pytest.approx(3 / 4)
    metrics.update_finetune_metrics_eval(top1, top5, logits[:1], labels[:1])
    assert top1.result() == pytest.approx(3 / 5)
    assert top1.totals().tolist() == [3.0, 5.0]


def test_jsonl_summary_writer(tmp_path):
    w = metrics.JsonlSummaryWriter(str(tmp_path / 'run'))
    m = metrics.Mean('train/contrast_loss')
    m.update_state(torch.tensor(2.0)); m.update_state(torch.tensor(4.0))
    metrics.log_and_write_metrics_to_summary([m], 30, w)
    w.scalar('learning_rate', 0.125, 30)
    w.close()
    lines = [json.loads(l) for l in open(os.path.join(str(tmp_path / 'run'), 'summaries.jsonl'))]
    assert lines == [{'step': 30, 'tag': 'train/contrast_loss', 'value': 3.0}, {'step': 30, 'tag': 'learning_rate', 'value': 0.125}]


def test_perform_evaluation_host_logic_with_a_stand_in_model(tmp_path):
    """tf2/run.py:348-432 without kernels: restore-after-first-batch, metric accumulation, result / flags / summary files."""
    from simclr_amd.flags import FLAGS
    from simclr_amd.run import perform_evaluation

    class _Sup:
        def __init__(self, z):
            self.z = z

        def dense(self):
            return self.z

    class _EvalToy(Layer):
        def __init__(self):
            self.scale = Variable('toy/scale:0', torch.ones(1), trainable=False)

        def __call__(self, features, training):
            assert training is False
            # logits = per-class mean intensity of the image, scaled by a checkpointed variable
            return None, _Sup(features.mean((1, 2)) * self.scale.value)

    FLAGS.reset()
    d = str(tmp_path / 'eval')
    src = _EvalToy()
    src.scale.value.fill_(-1.0)                     # the checkpoint flips the sign of every logit
    ck = Checkpoint(model=src)
    ck.global_step = 42
    path = CheckpointManager(ck, d).save(42)

    # three "classes" = the three colour channels; label = brightest channel -> after the restore (sign flip) the
    # model predicts the DARKEST channel, so top-1 is 0 and top-5 (k > classes) is 1
    g = torch.Generator().manual_seed(0)

    def batches():
        while True:
            x = torch.rand(4, 8, 8, 3, generator=g)
            lab = torch.nn.functional.one_hot(x.mean((1, 2)).argmax(1), 3).float()
            yield x, {'labels': lab}

    model = _EvalToy()
    result = perform_evaluation(model, batches(), 3, path, None, model_dir=d)
    assert float(model.scale.value) == -1.0
    assert result['global_step'] == 42
    assert result['eval/label_top_1_accuracy'] == 0.0 and result['eval/label_top_5_accuracy'] == 1.0
    for f in ('result.json', 'result_42.json', 'flags.json', 'summaries.jsonl'):
        assert os.path.exists(os.path.join(d, f)), f
    assert json.load(open(os.path.join(d, 'result.json')))['global_step'] == 42.0
    tags = [json.loads(l)['tag'] for l in open(os.path.join(d, 'summaries.jsonl'))]
    assert tags == ['eval/regularization_loss', 'eval/label_top_1_accuracy', 'eval/label_top_5_accuracy']
    flags = json.load(open(os.path.join(d, 'flags.json')))
    assert flags['temperature'] == 0.1 and flags['train_mode'] == 'pretrain'
    # pretraining without the linear-eval head: evaluation is skipped (run.py:350-352)
    FLAGS.update(lineareval_while_pretraining=False)
    assert perform_evaluation(model, batches(), 1, path, None, model_dir=d) is None
    FLAGS.reset()
