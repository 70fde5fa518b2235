"""Oracle: NT-Xent contrastive loss + cross-replica concat (numpy float64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED TO THE REFERENCE'S SOURCE (tests/golden/reference_pin.npz: tf2/*.py executed on oracle/tfshim.py); TensorFlow's own kernels unpinned.

Restates /root/reference/tf2/objective.py:35-127 and the consumers of its
outputs in /root/reference/tf2/metrics.py:23-36.
"""
import numpy as np

LARGE_NUM = 1e9  # tf2/objective.py:24


def l2_normalize(x, axis=-1, epsilon=1e-12):
    """tf.math.l2_normalize (tf2/objective.py:54): x * rsqrt(max(sum(x^2), eps))."""
    sq = np.sum(np.square(x), axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(sq, epsilon))


def _softmax_xent(labels_onehot, logits):
    """tf.nn.softmax_cross_entropy_with_logits (tf2/objective.py:83-86)."""
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    return lse - (labels_onehot * logits).sum(axis=1)


def tpu_cross_replica_concat(per_replica_tensors):
    """tf2/objective.py:92-127 over an explicit list of per-replica tensors.

    The reference scatters each replica's tensor into a zero [R, ...] tensor
    (:114-117), all-reduces with SUM (:121-122) and flattens the replica axis
    (:127).  With every replica's value present that is a concatenation in
    replica order.
    """
    r = len(per_replica_tensors)
    if r <= 1:  # :103-104
        return per_replica_tensors[0]
    ext = np.zeros((r,) + per_replica_tensors[0].shape, dtype=per_replica_tensors[0].dtype)
    for i, t in enumerate(per_replica_tensors):
        contrib = np.zeros_like(ext)
        contrib[i] = t          # scatter_nd :114-117
        ext = ext + contrib     # all_reduce SUM :121-122
    return ext.reshape((-1,) + ext.shape[2:])  # :127


def add_contrastive_loss(hidden, hidden_norm=True, temperature=1.0,
                         all_hiddens=None, replica_id=0):
    """tf2/objective.py:35-89 for ONE replica.

    hidden: [2n, D] local projection-head output of this replica.
    all_hiddens: None (strategy=None path, :69-73) or the list of every
        replica's [2n, D] `hidden` (the strategy path, :58-68).
    Returns (loss, logits_ab [n, N], labels [n, 2N]).
    """
    hidden = np.asarray(hidden, dtype=np.float64)
    if hidden_norm:
        hidden = l2_normalize(hidden, -1)                    # :53-54
    hidden1, hidden2 = np.split(hidden, 2, 0)                # :55
    batch_size = hidden1.shape[0]                            # :56
    if all_hiddens is not None:
        hs = [np.asarray(h, dtype=np.float64) for h in all_hiddens]
        if hidden_norm:
            hs = [l2_normalize(h, -1) for h in hs]
        hidden1_large = tpu_cross_replica_concat([np.split(h, 2, 0)[0] for h in hs])  # :60
        hidden2_large = tpu_cross_replica_concat([np.split(h, 2, 0)[1] for h in hs])  # :61
        enlarged = hidden1_large.shape[0]                    # :62
        labels_idx = np.arange(batch_size) + replica_id * batch_size   # :67
        labels = np.eye(enlarged * 2)[labels_idx]            # :68
        masks = np.eye(enlarged)[labels_idx]                 # :69
    else:
        hidden1_large, hidden2_large = hidden1, hidden2      # :71-72
        labels = np.eye(batch_size * 2)[np.arange(batch_size)]   # :73
        masks = np.eye(batch_size)[np.arange(batch_size)]        # :74
    logits_aa = hidden1 @ hidden1_large.T / temperature      # :76
    logits_aa = logits_aa - masks * LARGE_NUM                # :77
    logits_bb = hidden2 @ hidden2_large.T / temperature      # :78
    logits_bb = logits_bb - masks * LARGE_NUM                # :79
    logits_ab = hidden1 @ hidden2_large.T / temperature      # :80
    logits_ba = hidden2 @ hidden1_large.T / temperature      # :81
    loss_a = _softmax_xent(labels, np.concatenate([logits_ab, logits_aa], 1))  # :83-84
    loss_b = _softmax_xent(labels, np.concatenate([logits_ba, logits_bb], 1))  # :85-86
    loss = np.mean(loss_a + loss_b)                          # :87
    return loss, logits_ab, labels                           # :89


def contrastive_metrics(logits_ab, labels):
    """tf2/metrics.py:28-35: (contrast_acc, contrast_entropy)."""
    acc = np.mean((labels.argmax(1) == logits_ab.argmax(1)).astype(np.float32))   # :28-31
    m = logits_ab.max(axis=1, keepdims=True)
    p = np.exp(logits_ab - m)
    p = p / p.sum(axis=1, keepdims=True)                     # :33
    ent = -np.mean(np.sum(p * np.log(p + 1e-8), -1))         # :34-35
    return float(acc), float(ent)


def contrastive_loss_and_grad(all_hiddens, hidden_norm=True, temperature=1.0):
    """Loss and d(loss_total)/d(hidden_r) for every replica, float64, analytic.

    Emulates R replicas running tf2/run.py:577-622: each replica computes its
    local loss (objective.py:87), divides by R (run.py:617) and gradients are
    SUMmed across replicas by apply_gradients (run.py:622).  Because the
    cross-replica concat is differentiable (its transpose is an all-reduce
    SUM, objective.py:114-122) every replica's hidden receives gradient from
    every replica's loss.  Returns (per_replica_losses, grads) where grads[r]
    = d( sum_q loss_q / R ) / d hidden_r  with hidden_r the UN-normalised input.
    """
    R = len(all_hiddens)
    hs = [np.asarray(h, dtype=np.float64) for h in all_hiddens]
    n = hs[0].shape[0] // 2
    D = hs[0].shape[1]
    N = R * n
    if hidden_norm:
        inv = [1.0 / np.sqrt(np.maximum(np.sum(h * h, -1, keepdims=True), 1e-12)) for h in hs]
        zs = [h * i for h, i in zip(hs, inv)]
    else:
        zs = hs
    z1 = np.concatenate([z[:n] for z in zs], 0)   # [N, D] = hidden1_large
    z2 = np.concatenate([z[n:] for z in zs], 0)   # [N, D] = hidden2_large
    losses = []
    g1 = np.zeros_like(z1)
    g2 = np.zeros_like(z2)
    for r in range(R):
        rows = slice(r * n, (r + 1) * n)
        q1, q2 = z1[rows], z2[rows]
        idx = np.arange(n) + r * n
        mask = np.zeros((n, N)); mask[np.arange(n), idx] = 1.0
        laa = q1 @ z1.T / temperature - mask * LARGE_NUM
        lbb = q2 @ z2.T / temperature - mask * LARGE_NUM
        lab = q1 @ z2.T / temperature
        lba = q2 @ z1.T / temperature
        onehot = np.zeros((n, 2 * N)); onehot[np.arange(n), idx] = 1.0
        la = np.concatenate([lab, laa], 1)
        lb = np.concatenate([lba, lbb], 1)
        losses.append(np.mean(_softmax_xent(onehot, la) + _softmax_xent(onehot, lb)))

        def dsoft(l):
            m = l.max(1, keepdims=True)
            p = np.exp(l - m)
            p /= p.sum(1, keepdims=True)
            return (p - onehot) / n / R          # mean over n rows (:87), /R (run.py:617)
        da, db = dsoft(la), dsoft(lb)
        dab, daa = da[:, :N], da[:, N:]
        dba, dbb = db[:, :N], db[:, N:]
        # d wrt the local (query) rows
        g1[rows] += (dab @ z2 + daa @ z1) / temperature
        g2[rows] += (dba @ z1 + dbb @ z2) / temperature
        # d wrt the gathered (key) rows: transpose of the concat = SUM over replicas
        g1 += (daa.T @ q1 + dba.T @ q2) / temperature
        g2 += (dab.T @ q1 + dbb.T @ q2) / temperature
    grads = []
    for r in range(R):
        rows = slice(r * n, (r + 1) * n)
        gz = np.concatenate([g1[rows], g2[rows]], 0)
        if hidden_norm:
            z = zs[r]
            # d/dx of x * rsqrt(max(|x|^2, eps)) for |x|^2 > eps
            gh = (gz - z * np.sum(z * gz, -1, keepdims=True)) * inv[r]
        else:
            gh = gz
        grads.append(gh)
    return losses, grads


def replica_partials(all_hiddens, r, hidden_norm=True, temperature=1.0):
    """What ONE replica's loss kernel produces before any collective (float64):
    (loss_r, z_r, inv_r, dz_local [2n,D], dz_all [2N,D]) with dz_* = d(loss_r / R) / d(normalised
    rows): dz_local through the replica's own (query) rows, dz_all through the gathered rows in
    [z1_all; z2_all] order.  Summing dz_all over replicas and slicing the replica's slot is the
    transpose of tf2/objective.py:114-122 (all_reduce SUM); used to test simclr_amd/comm.py."""
    R = len(all_hiddens)
    hs = [np.asarray(h, dtype=np.float64) for h in all_hiddens]
    n = hs[0].shape[0] // 2
    N = R * n
    if hidden_norm:
        inv = [1.0 / np.sqrt(np.maximum(np.sum(h * h, -1, keepdims=True), 1e-12)) for h in hs]
        zs = [h * i for h, i in zip(hs, inv)]
    else:
        inv = [np.ones((2 * n, 1)) for _ in hs]
        zs = hs
    z1 = np.concatenate([z[:n] for z in zs], 0)
    z2 = np.concatenate([z[n:] for z in zs], 0)
    rows = slice(r * n, (r + 1) * n)
    q1, q2 = z1[rows], z2[rows]
    idx = np.arange(n) + r * n
    mask = np.zeros((n, N)); mask[np.arange(n), idx] = 1.0
    onehot = np.zeros((n, 2 * N)); onehot[np.arange(n), idx] = 1.0
    la = np.concatenate([q1 @ z2.T / temperature, q1 @ z1.T / temperature - mask * LARGE_NUM], 1)
    lb = np.concatenate([q2 @ z1.T / temperature, q2 @ z2.T / temperature - mask * LARGE_NUM], 1)
    loss = np.mean(_softmax_xent(onehot, la) + _softmax_xent(onehot, lb))

    def dsoft(l):
        m = l.max(1, keepdims=True)
        p = np.exp(l - m)
        p /= p.sum(1, keepdims=True)
        return (p - onehot) / n / R
    da, db = dsoft(la), dsoft(lb)
    dab, daa, dba, dbb = da[:, :N], da[:, N:], db[:, :N], db[:, N:]
    dq1 = (dab @ z2 + daa @ z1) / temperature
    dq2 = (dba @ z1 + dbb @ z2) / temperature
    dk1 = (daa.T @ q1 + dba.T @ q2) / temperature
    dk2 = (dab.T @ q1 + dbb.T @ q2) / temperature
    return loss, zs[r], inv[r], np.concatenate([dq1, dq2], 0), np.concatenate([dk1, dk2], 0)
