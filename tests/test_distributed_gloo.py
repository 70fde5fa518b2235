"""world_size-2/4 CPU tests (gloo) of the multi-replica path: simclr_amd/comm.py and GradSync.

The HIP kernels need a GPU, so here the per-replica kernel outputs are produced by the oracle
(float64) and everything BETWEEN the kernels -- the all-gather layout, the reduce-scatter transpose,
the rank-offset labels, the SyncBN statistic reduction and the bucketed gradient all-reduce -- is
the product code, run for real over torch.distributed.  The end result must equal the oracle's
global-batch gradient (R replicas == 1 replica on the global batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from oracle import ntxent as ont
        from simclr_amd import comm
        from simclr_amd.run import GradSync
        st = comm.Strategy()
        assert st.num_replicas_in_sync == world and st.rank == rank
        n, D, T = 6, 16, 0.1
        rng = np.random.default_rng(11)
        hs = [rng.standard_normal((2 * n, D)) for _ in range(world)]
        loss_r, z_r, inv_r, dz_local, dz_all = ont.replica_partials(hs, rank, True, T)
        # collective A forward: gathered block must be [z1_all; z2_all]
        z_all = comm.gather_hidden(torch.from_numpy(z_r), st)
        zs = [ont.replica_partials(hs, q, True, T)[1] for q in range(world)]
        ref_all = np.concatenate([z[:n] for z in zs] + [z[n:] for z in zs])
        assert np.abs(z_all.numpy() - ref_all).max() < 1e-15
        # tpu_cross_replica_concat mirror
        from simclr_amd.objective import tpu_cross_replica_concat
        cc = tpu_cross_replica_concat(torch.full((2, 3), float(rank)), st)
        assert cc.shape == (2 * world, 3) and all(float(cc[2 * i, 0]) == i for i in range(world))
        # collective A backward (transpose) + local part + l2norm backward == global-batch gradient
        slot = comm.scatter_hidden_grad(torch.from_numpy(dz_all), st).numpy()
        dz = dz_local + slot
        dh = (dz - z_r * np.sum(z_r * dz, -1, keepdims=True)) * inv_r
        _, grads = ont.contrastive_loss_and_grad(hs, True, T)
        assert np.abs(dh - grads[rank]).max() < 1e-14, np.abs(dh - grads[rank]).max()
        # collective C: statistic sums
        sums = torch.full((2, 5), float(rank + 1), dtype=torch.float64)
        st.all_reduce_sum(sums)
        assert float(sums[0, 0]) == world * (world + 1) / 2

        # collective B: bucketed gradient all-reduce covers the flat buffer exactly once
        class V:
            def __init__(self, name):
                self.name = name
        names = ['model/head_supervised/x', 'model/projection_head/y'] + \
                ['model/resnet/block_group%d/%s' % (g, c) for g in (4, 3, 2, 1) for c in 'ab'] + ['model/resnet/stem']
        class M:
            pass
        m = M()
        m._flat_order = [V(nm) for nm in names]
        m._flat_offsets = [64 * i for i in range(len(names))]
        m._flat_grads = torch.full((64 * len(names),), float(rank + 1))
        gs = GradSync(m, st)
        assert gs.ranges[0][0] == 0 and gs.ranges[-1][1] == m._flat_grads.numel()
        assert all(a[1] == b[0] for a, b in zip(gs.ranges[:-1], gs.ranges[1:]))
        for stage in (4, 3, 2, 1, 0):
            gs.on_stage(stage)
        gs.wait()
        assert torch.all(m._flat_grads == world * (world + 1) / 2)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, 'FAIL: ' + traceback.format_exc()))


@pytest.mark.parametrize('world', [2, 4])
def test_multi_replica_semantics_gloo(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in res), res


def _worker_forced(q, port):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', SIMCLR_FORCE_COLLECTIVES='1')
        dist.init_process_group('gloo', rank=0, world_size=1)
        from simclr_amd import comm
        st = comm.Strategy()
        # one replica, collectives forced on: three distinct communicators, every collective legal and the identity
        assert st.force and comm.collectives_on(st) and st.num_replicas_in_sync == 1
        assert st.stat_group is not st.grad_group and st.stat_group is not st.group
        z = torch.randn(12, 16)
        assert torch.equal(comm.gather_hidden(z, st), z)
        fin = comm.gather_hidden(z, st, async_op=True)
        assert torch.equal(fin(), z)
        assert torch.equal(comm.scatter_hidden_grad(z.clone(), st), z)
        assert torch.equal(comm.scatter_hidden_grad(z.clone(), st, async_op=True)(), z)
        s2 = torch.ones(2, 7, dtype=torch.float64)
        a, b = st.all_reduce_sum_many([s2.clone(), 2 * s2])
        assert torch.equal(a, s2) and torch.equal(b, 2 * s2)
        assert st.hidden_collectives == 4 and st.stat_collectives == 1
        os.environ.pop('SIMCLR_FORCE_COLLECTIVES')
        assert not comm.collectives_on(None)
        dist.destroy_process_group()
        q.put('ok')
    except Exception:  # noqa
        import traceback
        q.put('FAIL: ' + traceback.format_exc())


def test_forced_collectives_with_one_replica():
    """SIMCLR_FORCE_COLLECTIVES=1 (the switch the one-rank RCCL GPU test uses): the Strategy builds its three communicators
    and issues every collective with a single replica; shapes and results are those of the identity."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(q, _free_port()))
    p.start()
    r = q.get(timeout=300)
    p.join(timeout=60)
    assert r == 'ok', r
