"""Checkpoint / resume -- the logical content and cadence of /root/reference/tf2/run.py:308-337, 640-664.

The reference saves `tf.train.Checkpoint(model=model, global_step=optimizer.iterations, optimizer=optimizer)`
through a `tf.train.CheckpointManager(directory=FLAGS.model_dir, max_to_keep=FLAGS.keep_checkpoint_max)`
(run.py:310-315) and restores either the latest checkpoint of `model_dir` (everything) or
`FLAGS.checkpoint` (model weights only, run.py:320-335).  Here a checkpoint is one `torch.save`d file

    {'model': {variable name: fp32 tensor}, 'optimizer': {'iterations': int, 'slots': {variable name: tensor}},
     'global_step': int, 'format': 1}

keyed by the Keras-style variable names (`Variable.name`), so the content is layout-independent: the
compute copies (bf16 / transposed weights) are rebuilt from the fp32 masters after a restore.
Variables are created lazily at the first forward pass (like Keras layers), so a model must have been
called once before `restore`: run.main builds the variables with one inference-mode forward (no statistics or
weights change) and restores BEFORE step 0, so a resumed run repeats no step; perform_evaluation restores after
the first eval batch.  Under several replicas every rank restores, only replica 0 writes (`save(..., write=False)`
on the others keeps their bookkeeping in step without touching the directory).  `expect_partial` semantics: names that are absent on either side are reported in the
returned status, not fatal, unless `assert_consumed()` is asked for; a shape mismatch is always an error.
"""
import json
import os

import torch

from .resnet import RT

INDEX_NAME = 'checkpoint.json'       # stands in for the `checkpoint` text proto of tf.train.CheckpointManager


class RestoreStatus:
    def __init__(self, missing_in_ckpt, unused_in_ckpt, shape_mismatch):
        self.missing_in_checkpoint = missing_in_ckpt      # model variables the file does not hold
        self.unused_in_checkpoint = unused_in_ckpt        # file entries no variable asked for
        self.shape_mismatch = shape_mismatch

    def expect_partial(self):
        if self.shape_mismatch:
            raise ValueError('checkpoint tensors with a different shape: %s' % self.shape_mismatch[:5])
        return self

    def assert_consumed(self):
        self.expect_partial()
        if self.missing_in_checkpoint or self.unused_in_checkpoint:
            raise AssertionError('checkpoint not fully consumed: missing %s, unused %s' % (
                self.missing_in_checkpoint[:5], self.unused_in_checkpoint[:5]))
        return self


class Checkpoint:
    """tf.train.Checkpoint(model=, global_step=, optimizer=) stand-in.  `global_step` is read from /
    written to `optimizer.iterations` (the reference passes that very variable, run.py:515-516)."""

    def __init__(self, model=None, optimizer=None):
        self.model = model
        self.optimizer = optimizer
        self.global_step = 0

    # ---- save
    def state_dict(self):
        out = {'format': 1, 'model': {}, 'optimizer': None, 'global_step': int(self.global_step)}
        if self.model is not None:
            for v in self.model.variables:
                if v.name in out['model']:
                    raise ValueError('duplicate variable name %r' % v.name)
                out['model'][v.name] = v.value.detach().to('cpu', copy=True)
        if self.optimizer is not None:
            self.global_step = out['global_step'] = int(self.optimizer.iterations)
            slots = {}
            if self.model is not None:
                for v in self.model.variables:
                    s = self.optimizer._slots.get(id(v))
                    if s is not None:
                        slots[v.name] = s.detach().to('cpu', copy=True)
            out['optimizer'] = {'iterations': int(self.optimizer.iterations), 'slots': slots}
        return out

    def write(self, path):
        tmp = path + '.tmp'
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, path)           # readers never see a half-written file
        return path

    # ---- restore
    def restore(self, path, model_only=False):
        state = torch.load(path, map_location='cpu')
        if state.get('format') != 1:
            raise ValueError('%s is not a simclr_amd checkpoint' % path)
        return self._assign(state, model_only)

    def _assign(self, state, model_only):
        missing, mismatch = [], []
        used = set()
        variables = self.model.variables if self.model is not None else []
        for v in variables:
            t = state['model'].get(v.name)
            if t is None:
                missing.append(v.name)
                continue
            used.add(v.name)
            if tuple(t.shape) != tuple(v.value.shape):
                mismatch.append((v.name, tuple(t.shape), tuple(v.value.shape)))
                continue
            v.value.copy_(t.to(v.value.device))
        unused = [k for k in state['model'] if k not in used]
        if not model_only:
            self.global_step = int(state.get('global_step', 0))
            opt = state.get('optimizer')
            if self.optimizer is not None and opt is not None:
                self.optimizer.iterations = int(opt['iterations'])
                with_slot = [v for v in variables if v.name in opt['slots']]
                self.optimizer._create_slots(with_slot)
                for v in with_slot:
                    s = opt['slots'][v.name]
                    if tuple(s.shape) == tuple(v.value.shape):
                        self.optimizer._slots[id(v)].copy_(s.to(v.value.device))
                self.optimizer._key = None      # the LARS descriptor table is rebuilt on the next apply
        RT.weights_version += 1            # every compute copy (bf16 / transposed) is stale now
        return RestoreStatus(missing, unused, mismatch)


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory, max_to_keep): numbered files `ckpt-<n>.pt`, an index
    with the retained paths, oldest deleted beyond max_to_keep (None / 0 keeps everything)."""

    def __init__(self, checkpoint, directory, max_to_keep=5, checkpoint_name='ckpt'):
        self.checkpoint = checkpoint
        self.directory = directory
        self.max_to_keep = max_to_keep
        self.checkpoint_name = checkpoint_name
        self._paths = []
        idx = os.path.join(directory, INDEX_NAME) if directory else None
        if idx and os.path.exists(idx):
            with open(idx) as f:
                saved = json.load(f)
            self._paths = [p for p in saved.get('all_model_checkpoint_paths', [])
                           if os.path.exists(os.path.join(directory, p))]

    @property
    def checkpoints(self):
        return [os.path.join(self.directory, p) for p in self._paths]

    @property
    def latest_checkpoint(self):
        return self.checkpoints[-1] if self._paths else None

    def save(self, checkpoint_number=None, write=True):
        """write=False (replicas other than 0 of a multi-replica job): only record the new checkpoint in this
        manager's list, so that `latest_checkpoint` names the same file on every replica; replica 0 writes it."""
        if checkpoint_number is None:
            checkpoint_number = (self.checkpoint.optimizer.iterations if self.checkpoint.optimizer is not None
                                 else self.checkpoint.global_step)
        name = '%s-%d.pt' % (self.checkpoint_name, int(checkpoint_number))
        if write:
            os.makedirs(self.directory, exist_ok=True)
            self.checkpoint.write(os.path.join(self.directory, name))
        if name in self._paths:
            self._paths.remove(name)
        self._paths.append(name)
        if self.max_to_keep:
            while len(self._paths) > self.max_to_keep:
                old = self._paths.pop(0)
                if write:
                    try:
                        os.remove(os.path.join(self.directory, old))
                    except FileNotFoundError:
                        pass
        if write:
            tmp = os.path.join(self.directory, INDEX_NAME + '.tmp')
            with open(tmp, 'w') as f:
                json.dump({'model_checkpoint_path': self._paths[-1], 'all_model_checkpoint_paths': self._paths}, f)
            os.replace(tmp, os.path.join(self.directory, INDEX_NAME))
        return os.path.join(self.directory, name)

    def restore_or_initialize(self):
        if self.latest_checkpoint:
            return self.checkpoint.restore(self.latest_checkpoint)
        return None


def try_restore_from_checkpoint(model, optimizer, model_dir, checkpoint=None, keep_checkpoint_max=5,
                                zero_init_logits_layer=False):
    """tf2/run.py:308-337.  Latest checkpoint of `model_dir` (weights + step + optimizer state) if there is one,
    else `checkpoint` (weights only; optionally zero the supervised head, run.py:329-335).  The model must have been
    built (one forward pass) so that its variables exist.  Returns (manager, status-or-None)."""
    manager = CheckpointManager(Checkpoint(model=model, optimizer=optimizer), model_dir, max_to_keep=keep_checkpoint_max)
    status = None
    latest = manager.latest_checkpoint
    if latest:
        status = manager.checkpoint.restore(latest).expect_partial()
    elif checkpoint:
        status = Checkpoint(model=model).restore(checkpoint, model_only=True).expect_partial()
        if zero_init_logits_layer and getattr(model, 'supervised_head', None) is not None:
            for v in model.supervised_head.trainable_variables:
                v.value.zero_()
            RT.weights_version += 1
    return manager, status
