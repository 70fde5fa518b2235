"""The pretraining step and driver -- mirror of /root/reference/tf2/run.py for the hot path.

`make_single_step` restates `single_step` (tf2/run.py:557-622) line for line on the HIP
kernels: model forward, contrastive + supervised losses, metrics, weight decay, loss / R,
backward (hand-written, replaces tape.gradient :621), cross-replica gradient SUM (implicit in
apply_gradients :614-622; here a bucketed RCCL all-reduce overlapped with the rest of the
backward pass), LARS.  `main` accepts the reference's flags (simclr_amd/flags.py); the tfds
input pipeline, checkpoint manager, eval loop and SavedModel export (run.py:241-553) are out of
scope -- `--dataset=synthetic` feeds random two-view batches of the right shape.
"""
import json
import logging
import sys
import time

import torch
import torch.distributed as dist

from . import metrics, ops
from . import model as model_lib
from . import objective as obj_lib
from .comm import Strategy, collectives_on, num_replicas
from .flags import FLAGS
from .resnet import RT, join_wgrad_stream


def build_metrics():
    """The metric set of tf2/run.py:534-549."""
    m = {}
    names = ['train/weight_decay', 'train/total_loss']
    if FLAGS.train_mode == 'pretrain':
        names += ['train/contrast_loss', 'train/contrast_acc', 'train/contrast_entropy']
    if FLAGS.train_mode == 'finetune' or FLAGS.lineareval_while_pretraining:
        names += ['train/supervised_loss', 'train/supervised_acc']
    for n in names:
        m[n] = metrics.Mean(n)
    return m


class GradSync:
    """Collective B: SUM of every trainable gradient across replicas (tf2/run.py:614-622).

    Gradients live in ONE flat fp32 buffer ordered last-layer-first, cut into buckets at block
    group boundaries; each bucket's all-reduce is issued asynchronously as soon as the backward
    pass has produced it, so xGMI traffic overlaps the remaining dgrad/wgrad kernels."""

    def __init__(self, model, strategy):
        self.strategy = strategy
        self.flat = model._flat_grads
        order, offs = model._flat_order, model._flat_offsets
        marks = ['block_group4', 'block_group3', 'block_group2', 'block_group1']
        cuts = [0]
        for mk in marks:                      # bucket k ends where block group (4-k) ends
            last = max(i for i, v in enumerate(order) if mk in v.name)
            end = offs[last + 1] if last + 1 < len(order) else self.flat.numel()
            cuts.append(end)
        cuts.append(self.flat.numel())
        self.ranges = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        self.stage_to_bucket = {4: 0, 3: 1, 2: 2, 1: 3}
        self.works = []

    def on_stage(self, stage):
        """Called by the backward pass after block group `stage` (4..1) is done; 0 = stem done."""
        if not collectives_on(self.strategy):
            return
        join_wgrad_stream()          # this bucket's weight gradients may still be running on the side stream
        if stage == 0:
            idx = len(self.ranges) - 1 if len(self.ranges) > 4 else None
        else:
            idx = self.stage_to_bucket.get(stage)
        if idx is None or idx >= len(self.ranges):
            return
        a, b = self.ranges[idx]
        self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM,
                                          group=getattr(self.strategy, 'grad_group', self.strategy.group), async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def make_single_step(model, optimizer, strategy, all_metrics=None):
    """Returns single_step(features, labels) -- tf2/run.py:557-622."""
    m = all_metrics if all_metrics is not None else build_metrics()
    state = {'sync': None}
    RT.strategy = strategy

    def single_step(features, labels):
        ops.begin_step(features.device)
        projection_head_outputs, supervised_head_outputs = model(features, training=True)   # :577-578
        R = num_replicas(strategy)
        con_loss = sup_loss = None
        sup_box = {}

        def supervised_part():
            if supervised_head_outputs is not None:
                l = labels['labels'] if isinstance(labels, dict) else labels
                # labels are reused for both views (tf2/run.py:599-600: l = concat([l, l], 0))
                sup_box['loss'] = obj_lib.add_supervised_loss(labels=l, logits=supervised_head_outputs)   # :601
        if projection_head_outputs is not None:
            outputs = projection_head_outputs
            # collective A (all-gather of the hidden block) is in flight while the supervised loss is computed
            con_loss, logits_con, labels_con = obj_lib.add_contrastive_loss(                # :582-586
                outputs, hidden_norm=FLAGS.hidden_norm, temperature=FLAGS.temperature, strategy=strategy,
                overlap=supervised_part)
        else:
            supervised_part()
        sup_loss = sup_box.get('loss')
        weight_decay = model_lib.add_weight_decay(model, adjust_per_optimizer=True)         # :609-610

        # ---- backward of (loss / R): tf2/run.py:617-621 ----
        if model._flat_grads is None:
            model.allocate_flat_grads()
            state['sync'] = GradSync(model, strategy)
        sync = state['sync']
        model._wd_grad_scale = 1.0 / R
        # NT-Xent backward launches the reduce-scatter of the key-side gradient (transpose of collective A); the
        # supervised head's backward (independent: stop_gradient, tf2/model.py:276-277) runs while it is in flight
        if con_loss is not None:
            con_loss.backward_start(1.0 / R)
        d_sup = sup_loss.backward() if sup_loss is not None else None
        model.backward_supervised(d_sup)
        d_proj = con_loss.backward_finish() if con_loss is not None else None
        model.backward(d_proj, None, on_stage=sync.on_stage)
        join_wgrad_stream()
        sync.wait()
        optimizer.apply_gradients([(v.grad, v) for v in model._flat_order])                # :622
        RT.weights_version += 1
        ops.end_step()
        if strategy is not None:
            strategy.check_health()          # a timed-out statistics exchange of an earlier step raises here (no device sync)

        # ---- metrics (device scalars, no sync): tf2/run.py:587-613 -- update_pretrain_metrics_train,
        # update_finetune_metrics_train, weight_decay and total_loss = the sum of the loss terms, in two launches:
        # total first, then every running sum (metrics.Mean.attach / bump)
        if state.get('bank') is None or state['bank'].device != features.device:
            state['bank'] = torch.zeros(16, device=features.device, dtype=torch.float32)
            state['zero'] = torch.zeros(1, device=features.device, dtype=torch.float32)
            for i, name in enumerate(sorted(m)):
                m[name].attach(state['bank'], i)
        order = sorted(m)
        wd_t = weight_decay.reshape(-1)[:1] if torch.is_tensor(weight_decay) else state['zero']
        total = torch.empty(1, device=features.device, dtype=torch.float32)
        terms = [wd_t] + ([con_loss.value.reshape(-1)[:1]] if con_loss is not None else []) + \
                ([sup_loss.value.reshape(-1)[:1]] if sup_loss is not None else [])
        ops.accumulate_scalars(terms, total=total, total_mask=(1 << len(terms)) - 1)
        vals = {'train/weight_decay': wd_t, 'train/total_loss': total}
        if con_loss is not None:
            vals['train/contrast_loss'] = con_loss.value.reshape(-1)[:1]
            vals['train/contrast_acc'] = logits_con.contrast_acc.reshape(-1)[:1]
            vals['train/contrast_entropy'] = logits_con.contrast_entropy.reshape(-1)[:1]
        if sup_loss is not None:
            vals['train/supervised_loss'] = sup_loss.value.reshape(-1)[:1]
            vals['train/supervised_acc'] = sup_loss.acc.reshape(-1)[:1]
        names = [nm for nm in order if nm in vals]
        # one launch: bank[index(name)] += value(name); the metrics absent this step keep their sums
        srcs = [vals.get(nm, state['zero']) for nm in order]
        assert len(order) <= 16, 'the metric bank holds 16 running sums (simclr_accumulate_scalars)'
        # The loss / metric scalars above are views into the per-step arena, which the NEXT step zeroes and refills: the same
        # launch writes them into `keep` (a fresh tensor per step) and the handles this function returns are re-pointed at
        # it, so a caller may hold them across steps (deferred .item() logging, trajectory lists).
        keep = torch.empty(len(order), device=features.device, dtype=torch.float32)
        ops.accumulate_scalars(srcs, dst=state['bank'], copy=keep)
        for nm in names:
            m[nm].bump()
        at = {nm: keep[i:i + 1] for i, nm in enumerate(order) if nm in vals}
        if con_loss is not None and 'train/contrast_loss' in at:
            con_loss.value = at['train/contrast_loss']
            if 'train/contrast_acc' in at and 'train/contrast_entropy' in at:
                logits_con.keep(at['train/contrast_acc'], at['train/contrast_entropy'])
        if sup_loss is not None and 'train/supervised_loss' in at:
            sup_loss.value = at['train/supervised_loss']
            if 'train/supervised_acc' in at:
                sup_loss.acc = at['train/supervised_acc']
        if torch.is_tensor(weight_decay) and 'train/weight_decay' in at:
            weight_decay = at['train/weight_decay']
        return dict(con_loss=con_loss, sup_loss=sup_loss, weight_decay=weight_decay, total_loss=total,
                    logits_con=logits_con if con_loss is not None else None)

    single_step.metrics = m
    return single_step


def synthetic_batches(batch, image_size, num_classes, device, seed=0, pool=2):
    """i.i.d. U[0,1) two-view batches [b, H, W, 6] + one-hot labels."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    feats = [torch.rand(batch, image_size, image_size, 6, generator=g).to(device) for _ in range(pool)]
    labs = [torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes)
            .float().to(device) for _ in range(pool)]
    i = 0
    while True:
        yield feats[i % pool], {'labels': labs[i % pool]}
        i += 1


def init_distributed():
    """One process per GPU (torchrun): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env.
    SIMCLR_DIST_BACKEND=gloo: the ranks talk over gloo and, with SIMCLR_SHARE_GPU=1, all sit on cuda:0 -- the way the
    multi-rank launch path is exercised on a single-GPU box (RCCL refuses two ranks on one device).
    SIMCLR_FORCE_COLLECTIVES=1 with WORLD_SIZE=1: a one-rank RCCL communicator whose collectives are all issued."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force = os.environ.get('SIMCLR_FORCE_COLLECTIVES') == '1'
    if world <= 1 and not force:
        return None
    backend = os.environ.get('SIMCLR_DIST_BACKEND', 'nccl')
    share = os.environ.get('SIMCLR_SHARE_GPU') == '1'
    local = 0 if share else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        if world <= 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return Strategy()


def json_serializable(val):      # tf2/run.py:340-345
    try:
        json.dumps(val)
        return True
    except TypeError:
        return False


def synthetic_eval_batches(batch, image_size, num_classes, device, seed=0, pool=2):
    """Single-view eval batches [b, H, W, 3] + one-hot labels (the eval pipeline feeds one centre crop,
    tf2/data.py:52-62 with is_training=False)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    feats = [torch.rand(batch, image_size, image_size, 3, generator=g).to(device) for _ in range(pool)]
    labs = [torch.nn.functional.one_hot(torch.randint(0, num_classes, (batch,), generator=g), num_classes)
            .float().to(device) for _ in range(pool)]
    i = 0
    while True:
        yield feats[i % pool], {'labels': labs[i % pool]}
        i += 1


def _check_device_health(strategy):
    """Synchronise and raise if any in-kernel bounded wait of this process ever timed out: the peer-mapped SyncBN exchange
    (comm.PeerStats: NaN-poisoned statistics) or a split-tail partner of the persistent convolution grid (wrong tile)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if strategy is not None:
        strategy.check_health(wait=True)
    ops.check_split_tail_health()


def perform_evaluation(model, data, eval_steps, ckpt, strategy, model_dir=None):
    """tf2/run.py:348-432: restore `ckpt` (weights + global step), run `eval_steps` batches through
    model(features, training=False), accumulate eval/label_top_1_accuracy, eval/label_top_5_accuracy and
    eval/regularization_loss, write result.json, result_<step>.json and flags.json into model_dir.
    `data`: iterator of (features [b,H,W,3], {'labels': one-hot}).  Returns the result dict (None when skipped)."""
    from .checkpoint import Checkpoint
    import os
    if FLAGS.train_mode == 'pretrain' and not FLAGS.lineareval_while_pretraining:
        logging.info('Skipping eval during pretraining without linear eval.')          # :350-352
        return None
    regularization_loss = metrics.Mean('eval/regularization_loss')
    label_top_1_accuracy = metrics.Accuracy('eval/label_top_1_accuracy')
    label_top_5_accuracy = metrics.TopKCategoricalAccuracy(5, 'eval/label_top_5_accuracy')
    all_metrics = [regularization_loss, label_top_1_accuracy, label_top_5_accuracy]
    global_step = 0
    restored = False
    for i in range(eval_steps):
        features, labels = next(data)
        _, supervised_head_outputs = model(features, training=False)                 # :379
        assert supervised_head_outputs is not None
        if not restored and ckpt:
            # variables exist only after the first forward pass (lazy build): restore, then redo this batch
            logging.info('Restoring from %s', ckpt)
            c = Checkpoint(model=model)
            c.restore(ckpt, model_only=False).expect_partial()                       # :369-373
            global_step = c.global_step
            restored = True
            _, supervised_head_outputs = model(features, training=False)
        restored = True
        outputs = supervised_head_outputs.dense()
        l = labels['labels']
        metrics.update_finetune_metrics_eval(label_top_1_accuracy, label_top_5_accuracy, outputs, l)   # :383-384
        reg_loss = model_lib.add_weight_decay(model, adjust_per_optimizer=True)      # :385
        regularization_loss.update_state(reg_loss)
        logging.info('Completed eval for %d / %d steps', i + 1, eval_steps)
    # replicas evaluate disjoint shards: the accuracies are ratios of summed counts
    if strategy is not None and num_replicas(strategy) > 1:
        for m in (label_top_1_accuracy, label_top_5_accuracy):
            t = m.totals().to(RT.device)
            strategy.all_reduce_sum(t)
            m._hits, m._count = t[:1], float(t[1].item())
    writer = metrics.JsonlSummaryWriter(model_dir) if (model_dir and (strategy is None or strategy.rank == 0)) else None
    metrics.log_and_write_metrics_to_summary(all_metrics, global_step, writer)           # :400-403
    if writer is not None:
        writer.close()
    result = {m.name: float(m.result()) for m in all_metrics}
    result['global_step'] = int(global_step)
    logging.info(result)
    if model_dir and (strategy is None or strategy.rank == 0):
        os.makedirs(model_dir, exist_ok=True)
        for name in ('result.json', 'result_%d.json' % result['global_step']):       # :410-418
            with open(os.path.join(model_dir, name), 'w') as f:
                json.dump({k: float(v) for k, v in result.items()}, f)
        with open(os.path.join(model_dir, 'flags.json'), 'w') as f:                  # :419-427
            json.dump({k: v for k, v in FLAGS.flag_values_dict().items() if json_serializable(v)}, f)
    return result


def main(argv):
    """tf2/run.py:450-664 for --dataset=synthetic: train (with checkpoints every checkpoint_steps /
    checkpoint_epochs and resume from model_dir), eval, or train_then_eval."""
    from .checkpoint import try_restore_from_checkpoint
    import math
    FLAGS.parse(argv)
    logging.basicConfig(level=logging.INFO)
    strategy = init_distributed()
    R = num_replicas(strategy)
    rank0 = strategy is None or strategy.rank == 0
    if FLAGS.dataset != 'synthetic':
        raise NotImplementedError('only --dataset=synthetic is available offline (no tfds); '
                                  'feed real data through make_single_step from your own pipeline')
    num_classes = 10 if FLAGS.image_size <= 32 else 1000
    num_train_examples = 50000 if FLAGS.image_size <= 32 else 1281167
    num_eval_examples = 10000 if FLAGS.image_size <= 32 else 50000
    train_steps = model_lib.get_train_steps(num_train_examples)
    eval_steps = FLAGS.eval_steps or int(math.ceil(num_eval_examples / FLAGS.eval_batch_size))   # :476-478
    epoch_steps = int(round(num_train_examples / FLAGS.train_batch_size))                     # :479
    checkpoint_steps = FLAGS.checkpoint_steps or (FLAGS.checkpoint_epochs * epoch_steps)      # :486-487
    RT.reset()
    RT.strategy = strategy
    RT.device = torch.device('cuda', torch.cuda.current_device())
    model = model_lib.Model(num_classes)
    rep = 0 if strategy is None else strategy.rank

    if FLAGS.mode == 'eval':                                                             # :482-496 (one pass over
        from .checkpoint import CheckpointManager, Checkpoint                            #  the latest checkpoint)
        mgr = CheckpointManager(Checkpoint(model=model), FLAGS.model_dir, FLAGS.keep_checkpoint_max)
        ckpt = FLAGS.checkpoint or mgr.latest_checkpoint
        data = synthetic_eval_batches(FLAGS.eval_batch_size // R, FLAGS.image_size, num_classes, RT.device, seed=100 + rep)
        result = perform_evaluation(model, data, eval_steps, ckpt, strategy, FLAGS.model_dir)
        if rank0:
            print(json.dumps(result), flush=True)
        return result

    learning_rate = model_lib.WarmUpAndCosineDecay(FLAGS.learning_rate, num_train_examples)
    optimizer = model_lib.build_optimizer(learning_rate)
    step_fn = make_single_step(model, optimizer, strategy)
    per_replica = FLAGS.train_batch_size // R                                   # tf2/data.py:45
    data = synthetic_batches(per_replica, FLAGS.image_size, num_classes, RT.device, seed=rep)
    manager = None
    summary_writer = metrics.JsonlSummaryWriter(FLAGS.model_dir) if (FLAGS.model_dir and rank0) else None   # :526
    log_every = FLAGS.checkpoint_steps or 10
    step = 0
    if FLAGS.model_dir:
        # Build the variables with a forward-only pass in inference mode (no statistics or moving averages move), then
        # restore BEFORE step 0 as tf2/run.py:520-521 does: the latest checkpoint of model_dir (weights, BN moving
        # statistics, LARS slots, step) or, failing that, the weights of --checkpoint (slots stay zero, step 0).
        model(torch.zeros(2, FLAGS.image_size, FLAGS.image_size, 3, device=RT.device), training=False)
        manager, status = try_restore_from_checkpoint(model, optimizer, FLAGS.model_dir, FLAGS.checkpoint,
                                                      FLAGS.keep_checkpoint_max, FLAGS.zero_init_logits_layer)
        if status is not None and manager.latest_checkpoint:
            step = int(optimizer.iterations)
            logging.info('restored %s; continuing from step %d', manager.latest_checkpoint, step)
    t0 = time.time()
    while step < train_steps:
        features, labels = next(data)
        step_fn(features, labels)
        step += 1
        if step % log_every == 0:
            torch.cuda.synchronize()
            ops.check_split_tail_health()        # a split-tail partner that never arrived is an error, not a silent wrong tile
            dt = time.time() - t0
            t0 = time.time()
            if rank0:
                vals = {k: v.result() for k, v in step_fn.metrics.items()}
                vals.update(step=step, images_per_sec=FLAGS.train_batch_size * log_every / dt,
                            learning_rate=learning_rate(step))
                print(json.dumps(vals), flush=True)
                if summary_writer is not None:                                              # :645-652
                    metrics.log_and_write_metrics_to_summary(list(step_fn.metrics.values()), step, summary_writer)
                    summary_writer.scalar('learning_rate', learning_rate(step), step)
                    summary_writer.flush()
            for v in step_fn.metrics.values():
                v.reset_states()
        if manager is not None and (step % checkpoint_steps == 0 or step == train_steps):   # :640-648 (every steps_per_loop)
            # A timed-out statistics exchange poisons that step's statistics -- and the weights -- with NaN, and the per-step
            # health check only sees the PREVIOUS step's copy: synchronise and inspect the sticky device counters NOW, so that a
            # poisoned state is never written as `latest` (nor rotates a good checkpoint out).  Raises -> nothing is saved.
            _check_device_health(strategy)
            # every replica records the new checkpoint (same name everywhere), replica 0 alone writes it: otherwise
            # the other replicas' `latest_checkpoint` would still name the resume point in train_then_eval
            manager.save(step, write=rank0)
            if strategy is not None:
                dist.barrier()
    _check_device_health(strategy)       # time-outs of the last steps are reported too (the loop may end between two log intervals)
    if FLAGS.mode == 'train_then_eval' and manager is not None:                           # :657-660
        edata = synthetic_eval_batches(FLAGS.eval_batch_size // R, FLAGS.image_size, num_classes, RT.device, seed=100 + rep)
        result = perform_evaluation(model, edata, eval_steps, manager.latest_checkpoint, strategy, FLAGS.model_dir)
        if rank0:
            print(json.dumps(result), flush=True)
        return result
    return None


if __name__ == '__main__':
    main(sys.argv[1:])
