"""Replica strategy: one process per GPU over torch.distributed (backend 'nccl' = RCCL / xGMI).

Stands in for the `tf.distribute` strategy object the reference threads through
`add_contrastive_loss(..., strategy)` (tf2/objective.py:35-38) and `strategy.run`
(tf2/run.py:638).  The reference's collectives:
  A  tpu_cross_replica_concat = scatter-into-zeros + all_reduce SUM  (tf2/objective.py:92-127)
     -> here a true all_gather forward and reduce_scatter(SUM) backward (its transpose);
  B  gradient SUM inside apply_gradients (tf2/run.py:614-622) -> bucketed all_reduce;
  C  SyncBatchNormalization statistics (tf2/resnet.py:50-60) -> all_reduce of [2,C] fp64 sums.
Everything here is device-agnostic torch (works on CPU tensors with gloo), which is how the
multi-replica semantics are tested without GPUs (tests/test_distributed_gloo.py).
"""
import torch
import torch.distributed as dist


class Strategy:
    """Minimal replica context: num_replicas_in_sync, replica id and the collectives."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; use strategy=None for one replica')
        self.group = group
        self.num_replicas_in_sync = dist.get_world_size(group)
        self.replica_id_in_sync_group = dist.get_rank(group)

    @property
    def rank(self):
        return self.replica_id_in_sync_group

    # -- collective A forward: concat of every replica's tensor in replica order
    def all_gather_concat(self, tensor):
        R = self.num_replicas_in_sync
        out = torch.empty((R * tensor.shape[0],) + tuple(tensor.shape[1:]), device=tensor.device,
                          dtype=tensor.dtype)
        dist.all_gather_into_tensor(out, tensor.contiguous(), group=self.group)
        return out

    # -- collective A backward: SUM over replicas, keep this replica's slot
    def reduce_scatter_sum(self, tensor):
        R = self.num_replicas_in_sync
        n = tensor.shape[0] // R
        out = torch.empty((n,) + tuple(tensor.shape[1:]), device=tensor.device, dtype=tensor.dtype)
        if dist.get_backend(self.group) == 'nccl':
            dist.reduce_scatter_tensor(out, tensor.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
        else:  # gloo has no reduce_scatter: all_reduce + slice (same result)
            t = tensor.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(t[self.rank * n:(self.rank + 1) * n])
        return out

    # -- collectives B and C
    def all_reduce_sum(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor


def num_replicas(strategy):
    return 1 if strategy is None else strategy.num_replicas_in_sync


def replica_id(strategy):
    return 0 if strategy is None else strategy.replica_id_in_sync_group


def gather_hidden(z_local, strategy):
    """[z1_local; z2_local] ([2n,D]) -> [z1_all; z2_all] ([2N,D]): one all_gather of the fused
    block (both views together), then a re-layout from replica-major to view-major order."""
    R = num_replicas(strategy)
    if R <= 1:                               # tf2/objective.py:103-104
        return z_local
    n = z_local.shape[0] // 2
    g = strategy.all_gather_concat(z_local)  # [R*2n, D] = r0:[z1;z2], r1:[z1;z2], ...
    g = g.view(R, 2, n, -1).transpose(0, 1).reshape(2 * R * n, -1)
    return g.contiguous()


def scatter_hidden_grad(dz_all, strategy):
    """Transpose of gather_hidden: [2N,D] key-side gradient -> SUM over replicas of the rows that
    belong to this replica, as [2n,D] (= [dz1_slot; dz2_slot])."""
    R = num_replicas(strategy)
    if R <= 1:
        return dz_all
    n = dz_all.shape[0] // (2 * R)
    g = dz_all.view(2, R, n, -1).transpose(0, 1).reshape(R * 2 * n, -1).contiguous()
    return strategy.reduce_scatter_sum(g)    # [2n, D]
