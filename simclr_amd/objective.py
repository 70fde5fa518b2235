"""Contrastive loss functions -- drop-in mirror of /root/reference/tf2/objective.py.

`add_contrastive_loss(hidden, hidden_norm, temperature, strategy)` keeps the reference
signature and 3-tuple return (tf2/objective.py:35-89) but runs the fused NT-Xent HIP kernels:
l2-normalise -> RCCL all-gather of the hidden block (replacing the scatter + all_reduce of
`tpu_cross_replica_concat`, :92-127) -> tiled similarity + online masked softmax-CE on the
matrix cores.  The [n,N] logits and [n,2N] one-hot labels the reference returns are never
materialised on the hot path: `logits_con` / `labels_con` are lazy handles that also carry the
fused contrast accuracy / entropy (tf2/metrics.py:28-35); call `.dense()` for real tensors.
"""
import torch

from . import ops
from .flags import FLAGS
from .comm import collectives_on, gather_hidden, num_replicas, replica_id, scatter_hidden_grad
from .resnet import RT

LARGE_NUM = 1e9  # tf2/objective.py:24 (kept for reference; the kernel skips the masked column)


class _Loss:
    """A scalar loss living on the device, with the hand-written backward attached.
    backward() = backward_start() + backward_finish(): between the two the caller may enqueue independent work,
    which then overlaps with the collective the first half launched (the reduce-scatter of the key-side gradient)."""

    def __init__(self, value, backward_fn, start_fn=None, finish_fn=None):
        self.value = value            # 0-d / 1-element float32 device tensor
        self._backward = backward_fn
        self._start, self._finish = start_fn, finish_fn

    def backward(self, grad_scale=1.0):
        return self._backward(grad_scale)

    def backward_start(self, grad_scale=1.0):
        if self._start is not None:
            self._start(grad_scale)
        else:
            self._pending = self._backward(grad_scale)

    def backward_finish(self):
        if self._finish is not None:
            return self._finish()
        return self._pending

    def item(self):
        return float(self.value.item())

    __float__ = item


class LazyLogits:
    """logits_ab handle ([n, N], tf2/objective.py:80,89)."""

    def __init__(self, z_local, z_all, temperature, out, ensure_entropy):
        self._z_local, self._z_all, self._t = z_local, z_all, temperature
        self._out = out
        self._ensure_entropy = ensure_entropy
        n, N = z_local.shape[0] // 2, z_all.shape[0] // 2
        self.shape = (n, N)
        self._kept = None

    @property
    def contrast_acc(self):          # tf2/metrics.py:28-31
        return self._kept[0] if self._kept is not None else self._out[1]

    @property
    def contrast_entropy(self):      # tf2/metrics.py:33-35 (produced by the backward sweep)
        if self._kept is not None:
            return self._kept[1]
        self._ensure_entropy()
        return self._out[2]

    def keep(self, acc, entropy):
        """Re-point the two metrics at copies that outlive the step arena (run.make_single_step)."""
        self._kept = (acc, entropy)

    def dense(self):
        return ops.ntxent_logits_ab(self._z_local, self._z_all, self._t)


class LazyLabels:
    """one_hot(labels_idx, 2N) handle ([n, 2N], tf2/objective.py:67-68,73)."""

    def __init__(self, n, N, rank, device):
        self.n, self.N, self.rank, self.device = n, N, rank, device
        self.shape = (n, 2 * N)

    def dense(self):
        idx = torch.arange(self.n, device=self.device) + self.rank * self.n
        return torch.nn.functional.one_hot(idx, 2 * self.N).float()


def tpu_cross_replica_concat(tensor, strategy=None):
    """Reduce a concatenation of the `tensor` across replicas (tf2/objective.py:92-127).
    The reference builds it from scatter_nd + all_reduce(SUM); over RCCL it is an all_gather."""
    if not collectives_on(strategy):
        return tensor
    return strategy.all_gather_concat(tensor)


def add_contrastive_loss(hidden, hidden_norm=True, temperature=1.0, strategy=None, overlap=None):
    """Compute loss for model (tf2/objective.py:35-89).

    Args:
      hidden: float32 device tensor (bsz, dim) = [view-a rows; view-b rows].
      hidden_norm: whether or not to use normalization on the hidden vector.
      temperature: a `floating` number for temperature scaling.
      strategy: replica context (simclr_amd.comm.Strategy) or None.
      overlap: optional zero-argument callable run while the all-gather of the hidden block is in flight
        (collective A runs on the communicator's side stream; north_star: overlap it with independent work).
    Returns:
      A loss scalar (with .backward / .backward_start / .backward_finish), the logits handle, the labels handle.
    """
    assert hidden.dtype == torch.float32 and hidden.dim() == 2
    hidden = hidden.contiguous()
    if hidden_norm:
        z, inv = ops.l2norm_fwd(hidden)                          # :53-54
    else:
        z, inv = hidden, None
    n = z.shape[0] // 2
    R, rank = num_replicas(strategy), replica_id(strategy)
    pending = gather_hidden(z, strategy, async_op=True)          # :58-61 (collective A), asynchronous
    if overlap is not None:
        overlap()
    z_all = pending()
    # FLAGS.ntxent_matmul='f16x3' (opt-in): the sweeps' fp32 products as three fp16-piece MFMA terms -- l2-normalised rows only
    split = bool(hidden_norm) and getattr(FLAGS, 'ntxent_matmul', 'exact') == 'f16x3'
    out, row_stats, ws = ops.ntxent_fwd(z, z_all, rank, temperature, split=split)
    state = {'done': False, 'dz_local': None, 'slot': None}

    def backward_start(grad_scale=1.0):
        dz_local, dz_all = ops.ntxent_bwd(z, z_all, rank, temperature, row_stats, grad_scale, out, ws, split=split)
        state['done'] = True
        state['dz_local'] = dz_local
        state['slot'] = scatter_hidden_grad(dz_all, strategy, async_op=True)   # transpose of the concat, asynchronous

    def backward_finish():
        dz_local = state['dz_local']
        dz_slot = state['slot']()
        state['dz_local'] = state['slot'] = None
        ops.axpy_f32(1.0, dz_slot, dz_local)
        if hidden_norm:
            return ops.l2norm_bwd(z, inv, dz_local)
        return dz_local

    def backward(grad_scale=1.0):
        backward_start(grad_scale)
        return backward_finish()

    def ensure_entropy():
        if not state['done']:
            ops.ntxent_bwd(z, z_all, rank, temperature, row_stats, 0.0, out, ws, split=split)
            state['done'] = True

    loss = _Loss(out[0:1], backward, backward_start, backward_finish)
    logits_con = LazyLogits(z, z_all, temperature, out, ensure_entropy)
    labels_con = LazyLabels(n, n * R, rank, z.device)
    loss.normalized = z
    return loss, logits_con, labels_con


_CLASS_ID_CACHE = {}


def _class_ids(labels):
    """int32 class ids of a one-hot (or already integer) label tensor; cached per (storage, version), so a batch whose
    labels are reused (both views, several steps of a resident pool) is converted once."""
    key = (labels.data_ptr(), tuple(labels.shape), labels.dtype, labels._version)
    hit = _CLASS_ID_CACHE.get(key)
    if hit is not None and hit[0]() is labels:
        return hit[1]
    ids = labels.argmax(1) if labels.dim() == 2 else labels
    ids = ids.to(torch.int32).contiguous()
    if len(_CLASS_ID_CACHE) > 64:
        _CLASS_ID_CACHE.clear()
    import weakref
    _CLASS_ID_CACHE[key] = (weakref.ref(labels), ids)
    return ids


def add_supervised_loss(labels, logits):
    """Compute mean supervised loss over local batch (tf2/objective.py:27-32).

    labels: one-hot float [b or 2b, C] (as in the reference) or int class ids [b or 2b]; when
    it holds b rows and the logits 2b, the labels are reused for both views (tf2/run.py:599-600).
    logits: model.SupLogits.  Returns a loss scalar with .backward() -> dlogits and `.acc`.
    """
    labels = _class_ids(labels)
    out = ops.step_scalars(2, logits.z.device)
    gscale = 1.0 / num_replicas(RT.strategy)                    # loss / R, tf2/run.py:617
    dlogits = ops.bias_softmax_xent(logits.z, logits.bias, labels, logits.num_classes, gscale, out)
    loss = _Loss(out[0:1], lambda grad_scale=None: dlogits)
    loss.acc = out[1:2]
    return loss
