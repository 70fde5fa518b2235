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
import os

import torch
import torch.distributed as dist


class PeerUnavailable(RuntimeError):
    """Raised on EVERY rank (the decision is taken collectively) when the peer-mapped exchange cannot be set up or fails its
    self-test; the caller then keeps the collective library for collective C."""


class PeerStats:
    """Collective C over peer-mapped memory (csrc/comm.hip, simclr_comm_*): every rank's mailbox is mapped into every
    process through hipIpc, an all-reduce of <= max_doubles fp64 values is ONE single-workgroup launch per rank, summed in
    rank order (bit-identical on all replicas).  Default transport of collective C for world > 1 on the 'nccl' backend
    (SIMCLR_PEER_STATS=0 keeps RCCL; =1 forces it on, e.g. for gloo ranks sharing one GPU: tests/test_gpu_distributed.py).

    Failure handling:
      * set-up is agreed COLLECTIVELY: create / open results are exchanged with all_gather_object, and a short self-test
        (known values, checked on the host) runs before the first real exchange; if any rank fails any stage, every rank
        raises PeerUnavailable and the Strategy falls back to the collective library -- no rank is left waiting;
      * at run time a peer that misses the (bounded) wait turns the result of that exchange into NaN and bumps a STICKY
        device counter; `check_health()` -- called once per step by run.make_single_step -- reads the counter of the
        PREVIOUS step without a device synchronisation and raises."""

    SELF_TEST_EXCHANGES = 6

    def __init__(self, group, rank, world, device, max_doubles=16384):
        import ctypes
        from ._lib import lib
        L = lib()
        self.rank, self.world, self.max_doubles = rank, world, max_doubles
        torch.cuda.set_device(device)
        self._mailbox = None
        self._mapped = []
        # stage 1: create + export the local mailbox (local failure -> reported to everybody, nobody hangs)
        mailbox = ctypes.c_void_p()
        handle = (ctypes.c_ubyte * 64)()
        err = None
        try:
            if os.environ.get('SIMCLR_PEER_TEST_FAIL') == str(rank):      # test hook: this rank "cannot export its mailbox"
                raise RuntimeError('simulated comm_create failure (SIMCLR_PEER_TEST_FAIL)')
            L.comm_create(world, max_doubles, ctypes.byref(mailbox), handle)
            self._mailbox = mailbox.value
        except Exception as e:      # noqa: BLE001 -- any local failure must reach the collective decision below
            err = repr(e)
        import socket
        infos = [None] * world
        dist.all_gather_object(infos, (err, bytes(handle), socket.gethostname()), group=group)
        # hipIpc handles only mean something on the exporting host: ranks on different hosts keep the collective library
        hosts = {i[2] for i in infos}
        host_err = ('ranks span %d hosts: %s' % (len(hosts), sorted(hosts))) if len(hosts) > 1 else None
        self._agree([i[0] or host_err for i in infos], 'create')
        # stage 2: map every peer's mailbox
        self._peers = (ctypes.c_void_p * world)()
        err = None
        try:
            for r in range(world):
                if r == rank:
                    self._peers[r] = mailbox.value
                else:
                    mapped = ctypes.c_void_p()
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(infos[r][1])
                    L.comm_open(buf, ctypes.byref(mapped))
                    self._peers[r] = mapped.value
                    self._mapped.append(mapped.value)
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)       # doubles as the barrier: every mailbox is mapped everywhere
        self._agree(errs, 'open')
        self._seq = 0
        self.status = torch.zeros(1, dtype=torch.int32, device=device)     # STICKY count of missed peer arrivals (0 = healthy)
        self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_event = None
        self.exchanges = 0
        # stage 3: self-test -- rank r contributes (r + 1) * (i + 1) at index i; the sum is known in closed form
        err = None
        L.comm_set_timeout(float(os.environ.get('SIMCLR_PEER_SELFTEST_TIMEOUT_S', '20')))     # a mapping that does not deliver is found in seconds
        try:
            tri = world * (world + 1) // 2
            for k in range(self.SELF_TEST_EXCHANGES):
                n = [1, 130, min(4096, max_doubles), 2, min(2048, max_doubles), 777][k % 6]
                n = min(n, max_doubles)
                x = (torch.arange(1, n + 1, dtype=torch.float64, device=device) * float((rank + 1) * (k + 1)))
                self.all_reduce_sum(x)
                want = torch.arange(1, n + 1, dtype=torch.float64, device=device) * float(tri * (k + 1))
                if not bool(torch.equal(x, want)):
                    err = 'self-test exchange %d: wrong sum (missed arrivals: %d)' % (k, int(self.status.item()))
                    break
            self.exchanges = 0
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        L.comm_set_timeout(0.0)             # training-time bound: minutes (SIMCLR_PEER_STATS_TIMEOUT_S, default 600 s)
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        self._agree(errs, 'self-test')

    def _agree(self, errs, stage):
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            self.close()
            raise PeerUnavailable('peer-mapped statistics exchange unavailable (%s failed on rank %s): %s' % (stage, bad[0][0], bad[0][1]))

    def close(self):
        from ._lib import lib
        L = lib()
        for m in self._mapped:
            try:
                L.comm_close(m)
            except Exception:      # noqa: BLE001
                pass
        self._mapped = []
        if self._mailbox:
            try:
                L.comm_destroy(self._mailbox)
            except Exception:      # noqa: BLE001
                pass
            self._mailbox = None

    def usable(self, tensor):
        return tensor.is_cuda and tensor.dtype == torch.float64 and tensor.is_contiguous() and 0 < tensor.numel() <= self.max_doubles

    def all_reduce_sum(self, tensor):
        import ctypes
        from ._lib import lib
        self._seq += 1
        lib().comm_stats_allreduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), ctypes.cast(self._peers, ctypes.c_void_p),
                                   self.rank, self.world, self.max_doubles, self._seq, self.status.data_ptr(),
                                   torch.cuda.current_stream(tensor.device).cuda_stream)
        self.exchanges += 1
        return tensor

    def check_health(self, wait=False):
        """Once per step: raise if an exchange of an EARLIER step timed out.  No device synchronisation: the sticky counter is
        copied to pinned host memory asynchronously and the copy of the previous call is inspected (wait=True: synchronise and
        inspect the current value -- end of training, tests)."""
        if self._status_event is not None and (wait or self._status_event.query()):
            if wait:
                self._status_event.synchronize()
            missed = int(self._status_host[0])
            if missed:
                raise RuntimeError('SyncBatchNormalization statistics exchange: %d peer arrival(s) timed out on rank %d; the affected '
                                   'statistics were poisoned with NaN.  Re-run with SIMCLR_PEER_STATS=0 to use RCCL for collective C.'
                                   % (missed, self.rank))
            self._status_event = None
        if self._status_event is None:
            self._status_host.copy_(self.status, non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record()
        if wait:
            self._status_event.synchronize()
            missed = int(self._status_host[0])
            self._status_event = None
            if missed:
                raise RuntimeError('SyncBatchNormalization statistics exchange: %d peer arrival(s) timed out on rank %d' % (missed, self.rank))


class Strategy:
    """Minimal replica context: num_replicas_in_sync, replica id and the collectives.

    Three communicators, so that the three kinds of traffic never queue behind each other on one RCCL stream:
      group       -- collective A (hidden all-gather / reduce-scatter), latency-bound, on the loss critical path;
      stat_group  -- collective C (SyncBN [2,C] sums), ~100 tiny all-reduces per step on the critical path;
      grad_group  -- collective B (bucketed gradient all-reduce, tens of MB each), overlapped with the backward.
    With a single communicator every small statistic all-reduce of the backward pass waits for the bucket that was
    issued just before it."""

    def __init__(self, group=None, separate_groups=True):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; use strategy=None for one replica')
        self.group = group
        self.num_replicas_in_sync = dist.get_world_size(group)
        self.replica_id_in_sync_group = dist.get_rank(group)
        self.stat_group = self.grad_group = group
        # SIMCLR_FORCE_COLLECTIVES=1: issue every collective even with ONE replica (each is then the identity) -- the way
        # the RCCL code paths (all_gather_into_tensor, reduce_scatter_tensor, the three communicators, async works) are
        # executed on a single-GPU box: tests/test_gpu_distributed.py::test_rccl_collectives_with_one_rank
        self.force = os.environ.get('SIMCLR_FORCE_COLLECTIVES') == '1'
        if separate_groups and (self.num_replicas_in_sync > 1 or self.force):
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            self.stat_group = dist.new_group(ranks)        # collective: every replica constructs its Strategy
            self.grad_group = dist.new_group(ranks)
        self.stat_collectives = 0                          # counters (bench / tests): collectives issued so far
        self.hidden_collectives = 0
        # collective C over peer-mapped memory instead of the collective library (see PeerStats): default for world > 1 on the
        # 'nccl' backend (one GPU per rank over xGMI), SIMCLR_PEER_STATS=1 forces it (gloo ranks sharing a GPU), =0 disables it.
        # Whether it is used is decided collectively: a failed set-up / self-test on ANY rank leaves RCCL in place on ALL ranks.
        self.peer_stats = None
        self.peer_stats_fallback = None
        want = os.environ.get('SIMCLR_PEER_STATS')
        if want is None:
            want = '1' if dist.get_backend(group) == 'nccl' else '0'
        if want == '1' and 1 < self.num_replicas_in_sync <= 16 and torch.cuda.is_available():
            try:
                self.peer_stats = PeerStats(self.stat_group, self.replica_id_in_sync_group, self.num_replicas_in_sync,
                                            torch.device('cuda', torch.cuda.current_device()))
            except PeerUnavailable as e:
                self.peer_stats_fallback = str(e)

    @property
    def rank(self):
        return self.replica_id_in_sync_group

    # -- collective A forward: concat of every replica's tensor in replica order
    def all_gather_concat(self, tensor, async_op=False):
        """async_op: returns (out, work); the collective runs on the communicator's own stream and `work.wait()`
        makes the CURRENT stream wait for it -- whatever is enqueued in between overlaps with the transfer."""
        R = self.num_replicas_in_sync
        out = torch.empty((R * tensor.shape[0],) + tuple(tensor.shape[1:]), device=tensor.device,
                          dtype=tensor.dtype)
        work = dist.all_gather_into_tensor(out, tensor.contiguous(), group=self.group, async_op=async_op)
        self.hidden_collectives += 1
        return (out, work) if async_op else out

    # -- collective A backward: SUM over replicas, keep this replica's slot
    def reduce_scatter_sum(self, tensor, async_op=False):
        R = self.num_replicas_in_sync
        n = tensor.shape[0] // R
        out = torch.empty((n,) + tuple(tensor.shape[1:]), device=tensor.device, dtype=tensor.dtype)
        self.hidden_collectives += 1
        if dist.get_backend(self.group) == 'nccl':
            work = dist.reduce_scatter_tensor(out, tensor.contiguous(), op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=async_op)
            return (out, work) if async_op else out
        # gloo has no reduce_scatter: all_reduce + slice (same result)
        t = tensor.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(t[self.rank * n:(self.rank + 1) * n])
        return (out, None) if async_op else out

    # -- collective C (SyncBN statistics): its own communicator
    def all_reduce_sum(self, tensor):
        self.stat_collectives += 1
        if self.peer_stats is not None and self.peer_stats.usable(tensor):
            return self.peer_stats.all_reduce_sum(tensor)
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.stat_group)
        return tensor

    def close(self):
        """Release the peer-mapped mailbox and its hipIpc mappings (tests re-create strategies several times per process)."""
        ps, self.peer_stats = self.peer_stats, None
        if ps is not None:
            try:
                torch.cuda.synchronize()
            except Exception:      # noqa: BLE001
                pass
            ps.close()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    def check_health(self, wait=False):
        """Once per step (run.make_single_step): a timed-out peer-mapped exchange of an earlier step raises here."""
        if self.peer_stats is not None:
            self.peer_stats.check_health(wait=wait)

    def all_reduce_sum_many(self, tensors):
        """ONE all-reduce for several small tensors of one dtype (statistics of BatchNorms whose inputs do not
        depend on each other, e.g. a projection shortcut's BN and bn1 of the same block)."""
        if len(tensors) == 1:
            return [self.all_reduce_sum(tensors[0])]
        flat = torch.cat([t.reshape(-1) for t in tensors])
        self.all_reduce_sum(flat)
        out, o = [], 0
        for t in tensors:
            out.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
        return out


def num_replicas(strategy):
    return 1 if strategy is None else strategy.num_replicas_in_sync


def collectives_on(strategy):
    """True when the cross-replica collectives must be issued: several replicas, or one with SIMCLR_FORCE_COLLECTIVES."""
    return strategy is not None and (strategy.num_replicas_in_sync > 1 or getattr(strategy, 'force', False))


def replica_id(strategy):
    return 0 if strategy is None else strategy.replica_id_in_sync_group


def gather_hidden(z_local, strategy, async_op=False):
    """[z1_local; z2_local] ([2n,D]) -> [z1_all; z2_all] ([2N,D]): one all_gather of the fused
    block (both views together), then a re-layout from replica-major to view-major order.
    async_op: returns a zero-argument function that waits for the transfer and returns the gathered block."""
    R = num_replicas(strategy)
    if not collectives_on(strategy):         # tf2/objective.py:103-104
        return (lambda: z_local) if async_op else z_local
    n = z_local.shape[0] // 2

    def relayout(g):                         # [R*2n, D] = r0:[z1;z2], r1:[z1;z2], ...
        return g.view(R, 2, n, -1).transpose(0, 1).reshape(2 * R * n, -1).contiguous()
    if not async_op:
        return relayout(strategy.all_gather_concat(z_local))
    g, work = strategy.all_gather_concat(z_local, async_op=True)

    def finish():
        if work is not None:
            work.wait()
        return relayout(g)
    return finish


def scatter_hidden_grad(dz_all, strategy, async_op=False):
    """Transpose of gather_hidden: [2N,D] key-side gradient -> SUM over replicas of the rows that
    belong to this replica, as [2n,D] (= [dz1_slot; dz2_slot]).  async_op: returns a wait-and-get function."""
    R = num_replicas(strategy)
    if not collectives_on(strategy):
        return (lambda: dz_all) if async_op else dz_all
    n = dz_all.shape[0] // (2 * R)
    g = dz_all.view(2, R, n, -1).transpose(0, 1).reshape(R * 2 * n, -1).contiguous()
    if not async_op:
        return strategy.reduce_scatter_sum(g)    # [2n, D]
    out, work = strategy.reduce_scatter_sum(g, async_op=True)

    def finish():
        if work is not None:
            work.wait()
        return out
    return finish
