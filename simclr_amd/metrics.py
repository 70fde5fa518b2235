"""Training metrics -- mirror of /root/reference/tf2/metrics.py for the pretraining step.

`Mean` stands in for tf.keras.metrics.Mean (running mean, `update_state` / `result` /
`reset_states`); values stay on the device until `result()` so the step never syncs.
"""
import logging

import torch


class Mean:
    def __init__(self, name):
        self.name = name
        self._bank = None
        self.reset_states()

    def reset_states(self):
        self._sum = None
        self._count = 0
        if getattr(self, '_bank', None) is not None:
            self._bank[0][self._bank[1]] = 0.0

    def attach(self, sums, index):
        """Batched bookkeeping (run.make_single_step): this metric's running sum is element `index` of the device tensor
        `sums`, which ONE simclr_accumulate_scalars launch per step updates for all metrics; `bump()` counts the update."""
        if self._bank is not None and self._bank[0] is not sums:
            # re-attached to another step function's bank: carry the sum accumulated so far over instead of dropping it
            prev = self._bank[0][self._bank[1]].detach().reshape(1).clone()
            self._sum = prev if self._sum is None else self._sum + prev.to(self._sum.device)
        self._bank = (sums, index)

    def bump(self):
        self._count += 1

    def update_state(self, value):
        if hasattr(value, 'value'):
            value = value.value
        if not torch.is_tensor(value):
            value = torch.tensor(float(value))
        v = value.detach().reshape(-1)[:1].clone()
        self._sum = v if self._sum is None else self._sum + v.to(self._sum.device)
        self._count += 1

    def result(self):
        if self._count == 0:
            return 0.0
        tot = float(self._sum.item()) if self._sum is not None else 0.0
        if getattr(self, '_bank', None) is not None:
            tot += float(self._bank[0][self._bank[1]].item())
        return tot / self._count


class Accuracy:
    """tf.keras.metrics.Accuracy: running mean of (y_true == y_pred) over all samples seen."""

    def __init__(self, name):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self._hits = None
        self._count = 0

    def update_state(self, y_true, y_pred):
        h = (y_true.reshape(-1) == y_pred.reshape(-1)).sum().to(torch.float64).reshape(1)
        self._hits = h if self._hits is None else self._hits + h
        self._count += y_true.numel()

    def totals(self):
        """(hits, count) as a fp64 tensor -- what a multi-replica eval all-reduces."""
        dev = self._hits.device if self._hits is not None else 'cpu'
        h = self._hits if self._hits is not None else torch.zeros(1, dtype=torch.float64)
        return torch.cat([h.to(dev), torch.tensor([float(self._count)], dtype=torch.float64, device=dev)])

    def result(self):
        t = self.totals()
        return float(t[0].item()) / max(float(t[1].item()), 1.0)


class TopKCategoricalAccuracy(Accuracy):
    """tf.keras.metrics.TopKCategoricalAccuracy(k): one-hot labels, a sample counts when its target class is
    among the k largest predictions (ties resolved like tf.math.in_top_k: strictly-greater count < k)."""

    def __init__(self, k, name):
        self.k = k
        super().__init__(name)

    def update_state(self, y_true, y_pred):
        target = y_true.argmax(1)
        tv = y_pred.gather(1, target[:, None])
        hit = (y_pred > tv).sum(1) < self.k
        h = hit.sum().to(torch.float64).reshape(1)
        self._hits = h if self._hits is None else self._hits + h
        self._count += y_true.shape[0]


def update_finetune_metrics_eval(label_top_1_accuracy_metrics, label_top_5_accuracy_metrics, outputs, labels):
    """tf2/metrics.py:58-62.  outputs: dense logits [b, C]; labels: one-hot [b, C]."""
    label_top_1_accuracy_metrics.update_state(labels.argmax(1), outputs.argmax(1))
    label_top_5_accuracy_metrics.update_state(labels, outputs)


def update_pretrain_metrics_train(contrast_loss, contrast_acc, contrast_entropy, loss, logits_con, labels_con):
    """Updated pretraining metrics (tf2/metrics.py:23-36).  The accuracy (argmax(labels) ==
    argmax(logits_ab), :28-31) and the entropy of softmax(logits_ab) (:33-35) come fused out of
    the loss kernels when `logits_con` is the lazy handle."""
    contrast_loss.update_state(loss)
    if hasattr(logits_con, 'contrast_acc'):
        contrast_acc.update_state(logits_con.contrast_acc)
        contrast_entropy.update_state(logits_con.contrast_entropy)
        return
    acc = (labels_con.argmax(1) == logits_con.argmax(1)).float().mean()
    contrast_acc.update_state(acc)
    p = torch.softmax(logits_con, -1)
    contrast_entropy.update_state(-(p * torch.log(p + 1e-8)).sum(-1).mean())


def update_finetune_metrics_train(supervised_loss_metric, supervised_acc_metric, loss, labels, logits):
    """tf2/metrics.py:49-55."""
    supervised_loss_metric.update_state(loss)
    if hasattr(loss, 'acc'):
        supervised_acc_metric.update_state(loss.acc)
        return
    acc = (labels.argmax(1) == logits.argmax(1)).float().mean()
    supervised_acc_metric.update_state(acc)


class JsonlSummaryWriter:
    """Stand-in for tf.summary.create_file_writer(model_dir) (tf2/run.py:356, :526): every scalar becomes one JSON
    line {"step": s, "tag": name, "value": v} in <model_dir>/summaries.jsonl (TensorBoard-importable, grep-able).
    Callable as writer(name, value, step) -- the hook log_and_write_metrics_to_summary expects."""

    def __init__(self, model_dir, filename='summaries.jsonl'):
        import os
        os.makedirs(model_dir, exist_ok=True)
        self.path = os.path.join(model_dir, filename)
        self._f = open(self.path, 'a')

    def scalar(self, name, value, step):
        import json
        self._f.write(json.dumps({'step': int(step), 'tag': name, 'value': float(value)}) + '\n')

    __call__ = scalar

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _float_metric_value(metric):
    return float(metric.result())


def log_and_write_metrics_to_summary(all_metrics, global_step, writer=None):
    """tf2/metrics.py:70-74: log every metric; `writer` (optional) is a callable(name, value, step)."""
    for metric in all_metrics:
        metric_value = _float_metric_value(metric)
        logging.info('Step: [%d] %s = %f', global_step, metric.name, metric_value)
        if writer is not None:
            writer(metric.name, metric_value, global_step)
