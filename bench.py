#!/usr/bin/env python
"""Benchmark of the SimCLR pretraining step on MI355X (BASELINE.json metric: images/sec).

  python bench.py --gpus N --steps K --warmup W
  N>1 without WORLD_SIZE in the environment: bench.py re-launches itself as
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
  (one rank per GPU over RCCL); launched that way by a driver it just runs as one rank.

One "step" = one full pretraining step (tf2/run.py:557-622): two-view ResNet-50 1x forward +
backward at 224 px, projection head, NT-Xent, linear-eval head, LARS -- on a synthetic batch that
is resident in HBM before the timed region.  One "image" = one dataset image = two views
(BASELINE.md).  Per-GPU batch is fixed at 512 images (BASELINE.json configs[1] at N=1,
configs[2] = global 4096 at N=8) => weak scaling; `value` = global_batch * K / max-over-ranks time.

WHICH MODE IS THE HEADLINE (round 6): the top-level value / dtype / roofline belong to the fastest mode whose outputs meet north_star's
tolerances (loss 1e-3 relative, normalised embeddings 1e-5 absolute against the reference-source fixtures, measured in this run:
`north_star_met`, `parity`): fp32 storage and accumulation, every product as three 16-bit-piece MFMA terms (fp16 pieces forward, bf16
pieces backward) -- `--dtype f32 --f32_matmul f16x3_3`, the defaults.  The bf16-storage speed mode (narrower than the reference's fp32;
misses the tolerances) is measured beside it as `speed_mode`, the exact fp32-input MFMA as `f32_mode`, round 5's six-bf16-term forward
as `parity_mode_bf16x6`; `--dtype bf16` makes the speed mode the headline of a run.

Timing protocol: W un-timed warm-up steps, then EXACTLY K steps between barrier +
device-synchronize pairs (wall clock, max over ranks -> `value`).  Inside the timed loop only ONE HIP
event per step boundary is recorded (on the launch stream) -> `step_ms` p10 / median / p90.  The
per-launch HIP-event profiler (2 events per kernel launch) runs in SEPARATE instrumented steps right
after the timed region, same process, same data -> `roofline`, `families`, `kernels`, `ntxent`.

JSON line extras:
  roofline      -- dominant kernel family (most GPU time).  Split modes: SURVEY 8(d)'s MINIMUM bytes at 4 B / element against 8 TB/s and
                   the 16-bit MFMA work (fp32 products x terms) against 2.5 PFLOP/s, whichever costs more time is the binding roof
                   (`families` gives forward / data gradient / weight gradient separately).  bf16 / exact fp32: algorithmic FLOPs
                   (2*M*N*K per launch) and minimum bytes over the summed launch time, roof chosen by the arithmetic intensity; the
                   as-implemented byte count is under `impl_*`.  `traffic` = PMC bytes per launch, sampled by two rocprofv3 --pmc child
                   runs of this script in the same mode.
  ntxent        -- the fused NT-Xent forward+backward kernels (north_star's named kernel): us, algorithmic
                   GB/s, TFLOP/s and fraction of the fp32-input MFMA peak.
  speed_mode / f32_mode / parity_mode_bf16x6 -- the same step in the other modes (N = 1), same K / W.
  allgather     -- N>1: bandwidth of collective A (all-gather of the hidden block) and of the gradient all-reduce.
  cpu_baseline  -- the CPU oracle (torch-CPU restatement of the TF2 reference; TensorFlow is not installed)
                   timed on this box's host cores (rank 0, N=1 only) on BASELINE configs[0] (ResNet-18,
                   CIFAR 32x32, batch 256), plus a ResNet-50/224 sample and the un-fused NT-Xent / per-tensor
                   LARS restatements.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
FLOP_PER_IMAGE = 49.15e9    # 2 views x (fwd+dgrad+wgrad), encoder + head
# (depth, width, SK) -> FLOP per image at 224 px (SURVEY 8(d) / BASELINE.md section 3: cfg2/3, cfg4, cfg5)
FLOP_PER_IMAGE_BY_MODEL = {(50, 1, False): 49.15e9, (50, 2, True): 296.6e9, (152, 3, True): 1893.6e9}
HERE = os.path.dirname(os.path.abspath(__file__))


def _timed_steps(fn, budget_s, max_steps, warm=1, min_steps=1):
    for _ in range(warm):                            # warm-up (allocator, MKL / oneDNN primitives)
        fn()
    n, t0 = 0, time.time()
    while True:
        fn()
        n += 1
        if n >= min_steps and (time.time() - t0 > budget_s or n >= max_steps):
            break
    return n, time.time() - t0


def cpu_baseline():
    """CPU oracle timings on the host cores (>= 3 warm-up + >= 10 timed steps of configs[0]): about 30 s.
    SIMCLR_CPU_THREADS overrides the thread count (the 16 below is the measured optimum on the GPU host: profiles/r05_cpu_threads.json)."""
    from collections import OrderedDict
    import numpy as np
    from oracle import lars as olars
    from oracle import ntxent as ont
    from oracle.model_torch import Config, init_model, train_step
    torch.manual_seed(0)
    # 16 threads: on the 128-core GPU host the small CIFAR-sized convolutions run SLOWER with all cores (14.6 images/s
    # at 128 threads vs the 8-core build container's 20.4) -- thread oversubscription, not a property of the algorithm
    prev = torch.get_num_threads()
    cores = min(int(os.environ.get('SIMCLR_CPU_THREADS', '16')), os.cpu_count() or 1)
    torch.set_num_threads(cores)

    def model_step(depth, size, b, classes, budget, max_steps, warm=1, min_steps=1):
        cfg = Config(resnet_depth=depth, image_size=size, num_classes=classes)
        params, state = init_model(cfg, seed=2)
        momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        images = torch.rand(b, size, size, 6)
        labels = torch.nn.functional.one_hot(torch.randint(0, classes, (b,)), classes).float()
        n, dt = _timed_steps(lambda: train_step(cfg, params, state, momenta, images, labels, 0.1), budget, max_steps, warm, min_steps)
        return b * n / dt, n, dt

    v1, n1, t1 = model_step(18, 32, 256, 10, 20.0, 10, warm=3, min_steps=10)     # BASELINE configs[0]: 3 warm-up + 10 timed steps
    v2, n2, t2 = model_step(50, 224, 8, 1000, 6.0, 2)           # the benchmarked architecture, small batch
    out = dict(value=round(v1, 2), unit='images/s', cores=cores, kind='port', steps=n1, warmup=3,
               sample='3 warm-up + %d timed full train steps of BASELINE configs[0]: ResNet-18, CIFAR 32x32, batch 256, 1 replica, fp32 '
                      '(torch-CPU restatement of tf2/run.py:557-622; TensorFlow is not installed), %.1f s' % (n1, t1),
               resnet50_224=dict(value=round(v2, 3), unit='images/s', batch=8,
                                 sample='%d steps of the ResNet-50 1x @224 step at batch 8, %.1f s' % (n2, t2)))
    # un-fused NT-Xent (tf2/objective.py:76-87: four [n,N] matmuls, concat, two softmax-CE) fwd + grad, cfg2 size
    n = 512
    hs = [np.random.default_rng(3).standard_normal((2 * n, 128)).astype(np.float32)]
    k, dt = _timed_steps(lambda: ont.contrastive_loss_and_grad(hs, True, 0.1), 2.0, 20)
    out['ntxent_unfused'] = dict(us=round(dt / k * 1e6, 1), n=n, N=n, D=128,
                                 sample='%d x oracle/ntxent.py contrastive_loss_and_grad (numpy), %.1f s' % (k, dt))
    # per-tensor LARS (tf2/lars_optimizer.py:83-137), one numpy update per tensor over ResNet-50-sized tensors
    rng = np.random.default_rng(4)
    sizes = [(3, 3, 512, 512), (1, 1, 2048, 512), (1, 1, 512, 2048), (1, 1, 1024, 256), (3, 3, 256, 256), (2048, 1000), (2048,), (512,)]
    ts = [(('conv2d_%d/kernel:0' if len(s) > 1 else 'batch_normalization_%d/gamma:0') % i,
           (rng.standard_normal(s) * 0.05).astype(np.float32), (rng.standard_normal(s) * 1e-3).astype(np.float32),
           np.zeros(s, np.float32)) for i, s in enumerate(sizes)]
    nel = sum(t[1].size for t in ts)

    def lars_all():
        for name, w, g, m in ts:
            olars.lars_apply(name, w, g, m, 0.3, momentum=0.9, weight_decay=1e-6,
                             exclude_from_weight_decay=['batch_normalization', 'bias', 'head_supervised'])
    k, dt = _timed_steps(lars_all, 2.0, 20)
    torch.set_num_threads(prev)
    out['lars_per_tensor'] = dict(us=round(dt / k * 1e6, 1), elems=nel, gbps=round(28.0 * nel * k / dt / 1e9, 2),
                                  sample='%d x oracle/lars.py lars_apply over %d tensors (numpy float64), %.1f s' % (k, len(ts), dt))
    return out


def make_roofline(kernel, flops, min_bytes, impl_bytes, total_ms, launches, steps, mfma_peak_tflops):
    """Roofline entry of the dominant kernel family.  `flops` / `min_bytes`: algorithmic totals over
    `launches` launches that took `total_ms` (HIP events on the launch stream).  The binding roof follows from the
    arithmetic intensity: below the ridge (MFMA peak / HBM peak) => 'hbm', else 'mfma'."""
    sec = max(total_ms, 1e-9) * 1e-3
    tflops = flops / sec / 1e12
    gbps = min_bytes / sec / 1e9
    intensity = flops / max(min_bytes, 1.0)
    ridge = mfma_peak_tflops * 1e12 / (PEAK_HBM_GBPS * 1e9)
    hbm = intensity < ridge
    d = dict(bound='hbm' if hbm else 'mfma', kernel=kernel)
    if hbm:
        d.update(achieved=round(gbps, 1), peak=PEAK_HBM_GBPS, unit='GB/s', frac=round(gbps / PEAK_HBM_GBPS, 4))
    else:
        d.update(achieved=round(tflops, 2), peak=mfma_peak_tflops, unit='TFLOP/s', frac=round(tflops / mfma_peak_tflops, 4))
    # strict_frac: SURVEY 8(d) bytes only (= frac when HBM-bound); fused_operands_frac: the same time against the bytes
    # the launches move by design (residual / BN input / mask operands of the fused epilogues included)
    d.update(strict_frac=round((gbps / PEAK_HBM_GBPS) if hbm else (tflops / mfma_peak_tflops), 4),
             fused_operands_frac=round(impl_bytes / sec / 1e9 / PEAK_HBM_GBPS, 4) if hbm else None,
             traffic=None, traffic_source=None, traffic_measured_in_run=False,
             bytes_rule='SURVEY 8(d) minimum: (input + output + weights) * elt per launch, each read / written once',
             algorithmic_bytes_per_launch=round(min_bytes / max(launches, 1)),
             flops_per_launch_avg=flops / max(launches, 1),
             arithmetic_intensity_flop_per_byte=round(intensity, 1), ridge_flop_per_byte=round(ridge, 1),
             achieved_tflops=round(tflops, 2), mfma_frac=round(tflops / mfma_peak_tflops, 4),
             achieved_alg_gbps=round(gbps, 1), hbm_frac=round(gbps / PEAK_HBM_GBPS, 4),
             impl_bytes_per_launch=round(impl_bytes / max(launches, 1)),
             impl_gbps=round(impl_bytes / sec / 1e9, 1),
             avg_launch_us=round(total_ms * 1e3 / max(launches, 1), 2), launches_per_step=launches // max(steps, 1),
             ms_per_step=round(total_ms / max(steps, 1), 3), measured_over='%d instrumented steps after the timed region' % steps)
    return d


def relaunch_multi_gpu(args):
    """`python bench.py --gpus N` with no torchrun environment: spawn the N ranks ourselves."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def percentiles(ms):
    s = sorted(ms)
    if not s:
        return None
    pick = lambda q: s[min(len(s) - 1, int(q * (len(s) - 1) + 0.5))]
    return dict(p10=round(pick(0.1), 3), median=round(pick(0.5), 3), p90=round(pick(0.9), 3), n=len(s))


def build_step(args, dtype, strategy, world, rank, dev, f32_matmul=None):
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step, synthetic_batches
    global_batch = args.per_gpu_batch * world
    FLAGS.reset()
    FLAGS.update(resnet_depth=args.resnet_depth, width_multiplier=args.width_multiplier, image_size=args.image_size,
                 sk_ratio=args.sk_ratio, train_batch_size=global_batch, compute_dtype=dtype, use_blur=args.use_blur,
                 learning_rate=0.075, learning_rate_scaling='sqrt', weight_decay=1e-6,
                 temperature=0.1, hidden_norm=True, global_bn=True, lineareval_while_pretraining=True,
                 f32_matmul=f32_matmul or getattr(args, 'f32_matmul', 'exact'))
    RT.reset()
    RT.strategy = strategy
    RT.device = dev
    model = model_lib.Model(1000)
    schedule = model_lib.WarmUpAndCosineDecay(FLAGS.learning_rate, 1281167)
    optimizer = model_lib.build_optimizer(schedule)
    optimizer.iterations = 1000   # past step 0 so the warm-up LR is non-zero (weights really move)
    step_fn = make_single_step(model, optimizer, strategy)
    data = synthetic_batches(args.per_gpu_batch, args.image_size, 1000, dev, seed=rank)
    return step_fn, data, global_batch, model


def sample_pmc_traffic(family, extra_args=()):
    """HBM bytes per step of the dominant kernel family, measured in THIS run: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, never combined with trace domains: the rule of
    MI355X_MICROARCH.md), two steps each, parsed like tools/pmc_traffic.py (KB counters x 1024, FETCH x 2 for the 16 B/lane
    loads every streaming kernel of this library uses).  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    if shutil.which('rocprofv3') is None:
        return None
    tot = {}
    calls = defaultdict(int)
    try:
        with tempfile.TemporaryDirectory(dir='/tmp') as td:
            env = dict(os.environ, TMPDIR='/tmp')
            for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
                out = os.path.join(td, ctr)
                cmd = ['rocprofv3', '--pmc', ctr, '--output-format', 'csv', '-d', out, '-o', 'p', '--', sys.executable,
                       os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--no_cpu_baseline', '--no_kernel_events',
                       '--no_f32', '--no_pmc', '--no_parity'] + list(extra_args)
                subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
                acc = defaultdict(float)
                files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
                if not files:
                    return None
                for fpath in files:
                    with open(fpath) as f:
                        for r in csv.DictReader(f):
                            if r['Counter_Name'] == ctr:
                                acc[r['Kernel_Name']] += float(r['Counter_Value']) * 1024.0
                                if ctr == 'FETCH_SIZE':
                                    calls[r['Kernel_Name']] += 1
                tot[ctr] = acc
    except Exception:
        return None
    steps = 2.0                                     # 1 warm-up + 1 timed step per child run
    names = set(tot['FETCH_SIZE']) | set(tot['WRITE_SIZE'])
    fam = [n for n in names if family in n]
    side = [n for n in names if 'aug_' in n]         # the augmentation side measurement is not part of the step
    by = lambda ns: sum(2.0 * tot['FETCH_SIZE'].get(n, 0.0) + tot['WRITE_SIZE'].get(n, 0.0) for n in ns) / steps
    return dict(family_bytes_per_step=by(fam), kernel_dispatches_per_step=sum(calls[n] for n in fam) / steps,
                step_total_bytes=by([n for n in names if n not in side]))


def measured_parity(dev, modes):
    """The `parity` object, MEASURED IN THIS RUN: one product training step (simclr_amd.run.make_single_step) per mode on the two
    well-conditioned reference-source fixtures (tests/golden/reference_pin.npz `r18_img` / `r50_img`: /root/reference/tf2's own
    model.py / run.py executed on oracle/tfshim.py) -- variables and images re-derived here from tests/golden/recipe.py (pure numpy,
    by variable NAME; no oracle import), compared with the reference's contrastive loss, total loss and l2-normalised embeddings.
    north_star: loss <= 1e-3 relative, normalised embeddings <= 1e-5 absolute."""
    import numpy as np
    from simclr_amd import model as model_lib
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import make_single_step
    gdir = os.path.join(HERE, 'tests', 'golden')
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import recipe
    ref = np.load(os.path.join(gdir, 'reference_pin.npz'))
    names = ['contrast_loss', 'contrast_acc', 'contrast_entropy', 'supervised_loss', 'supervised_acc', 'weight_decay', 'total_loss']
    out = dict(measured_in_run=True, north_star=dict(loss_rel=1e-3, emb_abs=1e-5),
               fixtures="tests/golden/reference_pin.npz: outputs of the reference's own tf2/model.py + tf2/run.py:557-622 executed on a float64 "
                        "numpy stand-in for TensorFlow (TensorFlow's kernels themselves unpinned)", modes={})
    for label, dtype, matmul in modes:
        worst = dict(loss_rel=0.0, total_loss_rel=0.0, emb_abs=0.0)
        cases = {}
        for tag, c in sorted(recipe.IMG_CASES.items()):
            FLAGS.reset()
            FLAGS.update(use_blur=False, resnet_depth=c['depth'], image_size=c['size'], compute_dtype=dtype, f32_matmul=matmul,
                         train_batch_size=c['batch'], weight_decay=1e-4)
            RT.reset()
            RT.device = dev
            model = model_lib.Model(c['classes'])
            with torch.no_grad():
                model(torch.zeros(2, c['size'], c['size'], 6, device=dev), training=False)       # builds the variables
            for v in model.variables:
                val = recipe.variable_value(v.name[len('model/'):], v.value.double().cpu().numpy(), c['perturb'])
                v.value.copy_(torch.from_numpy(val).to(torch.float32).to(dev))
            RT.weights_version += 1
            step = make_single_step(model, model_lib.build_optimizer(0.1), None)
            images = torch.from_numpy(recipe.structured_images(c['batch'], c['size'], 2, c['seed'])).float().to(dev)
            labels = torch.from_numpy(recipe.one_hot_labels(c['batch'], c['classes'], c['seed'])).float().to(dev)
            o = step(images, {'labels': labels})
            torch.cuda.synchronize()
            want = dict(zip(names, ref['step_%s_R1_metrics' % tag]))
            proj = ref[tag + '_proj']
            zr = proj / np.sqrt(np.maximum((proj * proj).sum(1, keepdims=True), 1e-12))
            z = o['con_loss'].normalized.double().cpu().numpy()
            con = float(o['con_loss'].value.reshape(-1)[0])
            tot = float(o['total_loss'].reshape(-1)[0])
            e = dict(loss_rel=abs(con - want['contrast_loss']) / abs(want['contrast_loss']),
                     total_loss_rel=abs(tot - want['total_loss']) / abs(want['total_loss']),
                     emb_abs=float(np.abs(z - zr).max()))
            cases[tag] = {k: float('%.3e' % v) for k, v in e.items()}
            for k in worst:
                worst[k] = max(worst[k], e[k])
            del model, step, o
        out['modes'][label] = dict(dtype=dtype, f32_matmul=matmul if dtype == 'f32' else None,
                                   loss_rel=float('%.3e' % worst['loss_rel']), total_loss_rel=float('%.3e' % worst['total_loss_rel']),
                                   emb_abs=float('%.3e' % worst['emb_abs']),
                                   north_star_met=bool(worst['loss_rel'] <= 1e-3 and worst['total_loss_rel'] <= 1e-3 and worst['emb_abs'] <= 1e-5),
                                   cases=cases)
    FLAGS.reset()
    RT.reset()
    return out


def ntxent_loop_us(n, dev, iters=50):
    """The whole NT-Xent path of one step -- l2norm, forward (sweep + two finalize kernels), backward (two sweeps in one launch + combine),
    l2norm backward: SEVEN kernels -- as `iters` back-to-back repetitions between two HIP events (launch gaps included)."""
    from simclr_amd import ops
    h = torch.randn(2 * n, 128, device=dev)
    ws = None

    def once():
        nonlocal ws
        z, inv = ops.l2norm_fwd(h)
        out, rs, ws = ops.ntxent_fwd(z, z, 0, 0.1, ws)
        dl, da = ops.ntxent_bwd(z, z, 0, 0.1, rs, 1.0, out, ws)
        ops.l2norm_bwd(z, inv, dl)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def collective_bench(strategy, n, dev, flat_numel):
    """Collective A as the step issues it ([2n,128] fp32 per rank, tf2/objective.py:92-127) and one full-buffer gradient
    all-reduce (tf2/run.py:614-622): HIP events on the current stream, which waits for the communicator's stream."""
    R = strategy.num_replicas_in_sync
    z = torch.randn(2 * n, 128, device=dev)

    def timeit(fn, iters=30, warm=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    t_ag = timeit(lambda: strategy.all_gather_concat(z))
    contributed = z.numel() * 4
    out = dict(bytes_contributed=contributed, bytes_gathered=contributed * R, us=round(t_ag, 1),
               # per-GPU wire traffic of an all-gather: (R-1) peers' blocks received
               gbps=round((R - 1) * contributed / (t_ag * 1e-6) / 1e9, 2), ranks=R)
    # collective C as the step issues it: one [2, 2048] fp64 statistic block (tf2/resnet.py:50-60), over the collective
    # library and -- when SIMCLR_PEER_STATS=1 mapped the mailboxes -- over the one-launch peer-mapped exchange (csrc/comm.hip)
    st = torch.randn(2, 2048, device=dev, dtype=torch.float64)
    out['stat_exchange_us'] = dict(library=round(timeit(lambda: dist.all_reduce(st, group=strategy.stat_group)), 1))
    if getattr(strategy, 'peer_stats', None) is not None:
        out['stat_exchange_us']['peer_mapped'] = round(timeit(lambda: strategy.peer_stats.all_reduce_sum(st)), 1)
        out['stat_exchange_us']['peers_missing'] = int(strategy.peer_stats.status.item())
    g = torch.randn(flat_numel, device=dev)
    t_ar = timeit(lambda: dist.all_reduce(g, group=strategy.grad_group), iters=10, warm=2)
    out['grad_allreduce'] = dict(bytes=flat_numel * 4, us=round(t_ar, 1),
                                 busbw_gbps=round(2.0 * (R - 1) / R * flat_numel * 4 / (t_ar * 1e-6) / 1e9, 2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--per_gpu_batch', type=int, default=512)
    ap.add_argument('--dtype', default='f32', choices=['bf16', 'f32'],
                    help="storage type of the headline step; default f32 with --f32_matmul f16x3_3 = the fastest mode that meets north_star's "
                         "tolerances (the bf16 speed mode is measured beside it as `speed_mode`)")
    ap.add_argument('--f32_matmul', default='f16x3_3', choices=['exact', 'bf16x3', 'bf16x6', 'bf16x6_3', 'f16x3_3'],
                    help='--dtype f32 only: matrix arithmetic of the fp32 step (FLAGS.f32_matmul)')
    ap.add_argument('--resnet_depth', type=int, default=50)
    ap.add_argument('--image_size', type=int, default=224)
    ap.add_argument('--width_multiplier', type=int, default=1)
    ap.add_argument('--sk_ratio', type=float, default=0.0)
    ap.add_argument('--use_blur', action='store_true', help='include the on-device batch_random_blur (reference default)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_kernel_events', action='store_true', help='skip the instrumented steps (rocprof runs)')
    ap.add_argument('--no_f32', '--no_side', dest='no_f32', action='store_true', help='skip the side measurements of the other modes (bf16 speed mode, exact fp32, bf16x6_3)')
    ap.add_argument('--no_pmc', action='store_true', help='do not sample HBM traffic (two rocprofv3 --pmc child runs of this script)')
    ap.add_argument('--no_parity', action='store_true', help='skip the in-run parity measurement against the reference-source fixtures')
    ap.add_argument('--prof_steps', type=int, default=3)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo: the ranks talk over gloo and share cuda:0 when the box has fewer GPUs than ranks (exercises '
                         'the N > 1 launch path on a single-GPU box; RCCL refuses two ranks on one device)')
    args = ap.parse_args()
    if args.backend == 'gloo':
        os.environ['SIMCLR_DIST_BACKEND'] = 'gloo'
        if torch.cuda.device_count() < args.gpus:
            os.environ['SIMCLR_SHARE_GPU'] = '1'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(relaunch_multi_gpu(args))

    from simclr_amd import ops
    from simclr_amd.run import init_distributed

    strategy = init_distributed()
    world = 1 if strategy is None else strategy.num_replicas_in_sync
    rank = 0 if strategy is None else strategy.rank
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    dev = torch.device('cuda', torch.cuda.current_device())
    step_fn, data, global_batch, model = build_step(args, args.dtype, strategy, world, rank, dev)

    def sync():
        if strategy is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        f, l = next(data)
        step_fn(f, l)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.reset_peak_memory_stats()
    coll0 = (strategy.stat_collectives, strategy.hidden_collectives) if strategy is not None else (0, 0)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        f, l = next(data)
        step_fn(f, l)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if strategy is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = global_batch * args.steps / elapsed
    metrics = {k: v.result() for k, v in step_fn.metrics.items()}
    peak_hbm_gb = torch.cuda.max_memory_allocated() / 1e9          # every device buffer of the step is a torch allocation
    coll_counts = None
    if strategy is not None:
        coll_counts = dict(stat_collectives_per_step=(strategy.stat_collectives - coll0[0]) / args.steps,
                           hidden_collectives_per_step=(strategy.hidden_collectives - coll0[1]) / args.steps)

    # ---- instrumented steps (per-launch HIP events), outside the headline region
    prof = None
    if not args.no_kernel_events and args.prof_steps > 0:
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        for _ in range(args.prof_steps):
            f, l = next(data)
            step_fn(f, l)
        sync()
        ops.PROFILER = None

    coll = None
    if strategy is not None:
        coll = collective_bench(strategy, args.per_gpu_batch, dev, int(model._flat_grads.numel()))
        # proof that N ranks took part: an all-reduce of ones over the collective library, and the set of devices the ranks sit on
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        devs = [None] * world
        dist.all_gather_object(devs, '%s:%d' % (socket.gethostname(), torch.cuda.current_device()))
        coll.update(coll_counts, backend=dist.get_backend(), rccl_ranks_seen=int(ones.item()), rank_devices=devs,
                    stat_transport='peer-mapped mailboxes (csrc/comm.hip)' if getattr(strategy, 'peer_stats', None) is not None
                    else 'collective library', stat_transport_fallback=getattr(strategy, 'peer_stats_fallback', None))
    if rank != 0:
        return

    default_cfg = (args.resnet_depth == 50 and args.image_size == 224 and args.width_multiplier == 1 and
                   args.sk_ratio == 0 and args.per_gpu_batch == 512)
    flop_img = FLOP_PER_IMAGE_BY_MODEL.get((args.resnet_depth, args.width_multiplier, args.sk_ratio > 0)) if args.image_size == 224 else None
    from simclr_amd import ops as _o

    def mode_label(dtype, matmul):
        return 'bf16' if dtype == 'bf16' else 'f32/' + matmul

    def mfma_terms(t):
        """16-bit-piece MFMA products per fp32 product: 3 / 6 bf16 terms, 13 = three fp16 terms, 0 = the exact fp32-input MFMA (1/16 rate)."""
        return 3 if t == 13 else (t or 16)

    def step_mfma_frac(v, dtype, matmul):
        """whole-step matrix work of the mode against its MFMA peak: bf16 = FLOPs / 2.5 PF; split modes = 16-bit MFMA work (forward third x
        forward terms + two backward thirds x backward terms) / 2.5 PF; exact fp32 = FLOPs / the fp32-input MFMA peak."""
        if not flop_img:
            return None
        if dtype == 'bf16':
            return round(v * flop_img / (world * PEAK_BF16_TFLOPS * 1e12), 4)
        tf, tb = _o.F32_MATMUL_TERMS[matmul]
        if tf == 0 and tb == 0:
            return round(v * flop_img / (world * PEAK_F32_TFLOPS * 1e12), 4)
        return round(v * flop_img * (mfma_terms(tf) + 2.0 * mfma_terms(tb)) / 3.0 / (world * PEAK_BF16_TFLOPS * 1e12), 4)

    def kernel_table(summ, P):
        return {fam: dict(launches_per_step=d['launches'] // P, ms_per_step=round(d['ms'] / P, 3),
                          tflops=round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2) if d['ms'] > 0 else None,
                          alg_gbps=round(d['bytes'] / (d['ms'] * 1e-3) / 1e9, 1) if d['ms'] > 0 else None) for fam, d in summ.items()}

    def plain_roofline(summ, P, dtype, pmc_args=None):
        """bf16 storage / exact fp32: dominant family = most GPU time (the fwd and dgrad launches are the same kernel template), algorithmic
        FLOPs and SURVEY 8(d) minimum bytes over the summed launch time, PMC traffic of the family sampled by two child runs."""
        fams = {'conv_igemm': [k for k in summ if k.startswith('conv_igemm')], 'conv_wgrad': ['conv_wgrad']}
        best, best_ms = None, -1.0
        for name, members in fams.items():
            ms = sum(summ[m]['ms'] for m in members if m in summ)
            if ms > best_ms:
                best, best_ms = name, ms
        members = [m for m in fams[best] if m in summ]
        fl = sum(summ[m]['flops'] for m in members)
        nl = sum(summ[m]['launches'] for m in members)
        by = sum(summ[m]['bytes'] for m in members)
        impl = sum(summ[m]['impl_bytes'] for m in members)
        r = make_roofline(best, fl, by, impl, best_ms, nl, P, PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_F32_TFLOPS)
        sampled = sample_pmc_traffic(best, extra_args=pmc_args) if pmc_args is not None else None
        if sampled:
            per_launch = sampled['family_bytes_per_step'] / max(r['launches_per_step'], 1)
            r.update(traffic=round(per_launch), traffic_measured_in_run=True,
                     traffic_source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this script in this mode on this box '
                                    '(separate passes, FETCH x2 for 16 B/lane loads, MI355X_MICROARCH.md)',
                     traffic_kernel_dispatches_per_step=sampled['kernel_dispatches_per_step'],
                     traffic_over_algorithmic=round(per_launch / r['algorithmic_bytes_per_launch'], 3),
                     step_total_traffic_gb=round(sampled['step_total_bytes'] / 1e9, 2))
        return r

    def split_roofline(sm, P, matmul, pmc_args=None):
        """fp32 storage / split 16-bit-piece matrix arithmetic.  Bytes: SURVEY 8(d)'s minimum at 4 B / element.  Matrix work: every fp32
        product runs as `terms` 16-bit MFMAs (forward: 3 fp16 or 6 bf16; backward: 3 bf16), so the MFMA roof of a family is 2.5 PFLOP/s /
        terms; the binding roof of a family is whichever limit costs more time."""
        tf, tb = _o.F32_MATMUL_TERMS[matmul]
        fam_terms = {'conv_igemm_fwd': mfma_terms(tf), 'conv_igemm_dgrad': mfma_terms(tb), 'conv_wgrad': mfma_terms(tb)}
        fams = {}
        for k, t in fam_terms.items():
            if k not in sm or sm[k]['ms'] <= 0:
                continue
            d = sm[k]
            sec = d['ms'] * 1e-3
            t_mfma = d['flops'] * t / (PEAK_BF16_TFLOPS * 1e12)          # seconds at the dense 16-bit MFMA peak
            t_hbm = d['bytes'] / (PEAK_HBM_GBPS * 1e9)                   # seconds at the HBM peak
            fams[k] = dict(launches_per_step=d['launches'] // P, ms_per_step=round(d['ms'] / P, 3), mfma_terms=t,
                           bound='mfma' if t_mfma >= t_hbm else 'hbm', frac=round(max(t_mfma, t_hbm) / sec, 4),
                           mfma_frac=round(t_mfma / sec, 4), hbm_frac=round(t_hbm / sec, 4),
                           fp32_product_tflops=round(d['flops'] / sec / 1e12, 2), mfma_work_tflops=round(d['flops'] * t / sec / 1e12, 2),
                           alg_gbps=round(d['bytes'] / sec / 1e9, 1), algorithmic_bytes_per_launch=round(d['bytes'] / max(d['launches'], 1)))
        # dominant family by time: the forward and data-gradient launches are ONE kernel template (conv_igemm_persistent)
        ig = [k for k in ('conv_igemm_fwd', 'conv_igemm_dgrad') if k in fams]
        ig_ms = sum(sm[k]['ms'] for k in ig)
        wg_ms = sm['conv_wgrad']['ms'] if 'conv_wgrad' in fams else -1.0
        members = ig if ig_ms >= wg_ms else ['conv_wgrad']
        name = 'conv_igemm' if ig_ms >= wg_ms else 'conv_wgrad'
        sec = sum(sm[k]['ms'] for k in members) * 1e-3
        fl = sum(sm[k]['flops'] for k in members)
        by = sum(sm[k]['bytes'] for k in members)
        nl = sum(sm[k]['launches'] for k in members)
        t_mfma = sum(sm[k]['flops'] * fam_terms[k] for k in members) / (PEAK_BF16_TFLOPS * 1e12)
        t_hbm = by / (PEAK_HBM_GBPS * 1e9)
        hbm = t_hbm > t_mfma
        r = dict(bound='hbm' if hbm else 'mfma', kernel=name,
                 achieved=round(by / sec / 1e9, 1) if hbm else round(t_mfma * PEAK_BF16_TFLOPS / sec, 2),
                 peak=PEAK_HBM_GBPS if hbm else PEAK_BF16_TFLOPS, unit='GB/s' if hbm else 'TFLOP/s',
                 unit_note=None if hbm else '16-bit MFMA work: fp32 products x terms (fp16 / bf16 pieces, dense peak 2.5 PFLOP/s either way)',
                 frac=round(max(t_mfma, t_hbm) / sec, 4), mfma_frac=round(t_mfma / sec, 4), hbm_frac=round(t_hbm / sec, 4),
                 traffic=None, traffic_measured_in_run=False,
                 bytes_rule='SURVEY 8(d) minimum at 4 B / element: (input + output + weights) per launch, each read / written once',
                 algorithmic_bytes_per_launch=round(by / max(nl, 1)), flops_per_launch_avg=fl / max(nl, 1),
                 fp32_product_tflops=round(fl / sec / 1e12, 2), avg_launch_us=round(sec * 1e6 / max(nl, 1), 2),
                 launches_per_step=nl // P, ms_per_step=round(sec * 1e3 / P, 3),
                 measured_over='%d instrumented steps after the timed region of this mode' % P)
        if pmc_args is not None:
            sampled = sample_pmc_traffic(name, extra_args=pmc_args)
            if sampled:
                per_launch = sampled['family_bytes_per_step'] / max(r['launches_per_step'], 1)
                r.update(traffic=round(per_launch), traffic_measured_in_run=True,
                         traffic_source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this script in this mode (separate passes, FETCH x2)',
                         traffic_kernel_dispatches_per_step=sampled['kernel_dispatches_per_step'],
                         traffic_over_algorithmic=round(per_launch / r['algorithmic_bytes_per_launch'], 3),
                         step_total_traffic_gb=round(sampled['step_total_bytes'] / 1e9, 2))
        other = {k: dict(launches_per_step=d['launches'] // P, ms_per_step=round(d['ms'] / P, 3)) for k, d in sm.items() if k not in fams}
        return r, fams, other

    def mode_roofline(summ, P, dtype, matmul):
        pmc_args = ('--dtype', dtype, '--f32_matmul', matmul) if (default_cfg and world == 1 and not args.no_pmc) else None
        if dtype == 'f32' and matmul != 'exact':
            return split_roofline(summ, P, matmul, pmc_args)
        return plain_roofline(summ, P, dtype, pmc_args), None, None

    roofline, families, other_kernels, kernels, ntx = None, None, None, {}, None
    if prof is not None:
        summ = prof.summary()
        P = args.prof_steps
        kernels = kernel_table(summ, P)
        roofline, families, other_kernels = mode_roofline(summ, P, args.dtype, args.f32_matmul)
        if 'ntxent_fwd' in summ and 'ntxent_bwd' in summ:
            # every kernel of the NT-Xent path: the two library calls (3 + 2 kernels) AND the two l2norm kernels around them
            parts = ['ntxent_fwd', 'ntxent_bwd', 'l2norm_fwd', 'l2norm_bwd']
            ms = sum(summ[k]['ms'] for k in parts if k in summ) / P
            nfl = (summ['ntxent_fwd']['flops'] + summ['ntxent_bwd']['flops']) / P
            nby = summ['ntxent_bwd']['bytes'] / P          # fused fwd+bwd I/O: 2*(2n+2N)*D*4
            loop_us = ntxent_loop_us(args.per_gpu_batch, dev) if world == 1 else None
            t_us = loop_us if loop_us else ms * 1e3
            ntx = dict(us=round(t_us, 1), us_events_in_step=round(ms * 1e3, 1),
                       us_by_call={k: round(summ[k]['ms'] / P * 1e3, 1) for k in parts if k in summ},
                       kernels=7, library_calls=sum(summ[k]['launches'] for k in parts if k in summ) // P,
                       kernel_list='l2norm_fwd, ntxent_fwd_partial, ntxent_finalize_rows, ntxent_reduce_out, ntxent_bwd_sweeps, ntxent_combine_all, l2norm_bwd',
                       alg_gbps=round(nby / (t_us * 1e-6) / 1e9, 2), tflops=round(nfl / (t_us * 1e-6) / 1e12, 2),
                       frac_f32_mfma=round(nfl / (t_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS, 4),
                       timing='`us` = 50 back-to-back repetitions of the seven kernels between two HIP events (launch gaps included); '
                              '`us_events_in_step` = per-call HIP events inside the instrumented training steps',
                       note='bound by the fp32-input matrix pipe, not HBM (AI 384-683 FLOP/B); HBM floor 1.5 us')

    # ---- the OTHER modes of the same step, same K / W, one GPU: `speed_mode` = bf16 storage (narrower than the reference's fp32: misses
    # north_star's tolerances, reported not credited), `f32_mode` = exact fp32-input MFMA (1/16 of the 16-bit rate), `parity_mode_bf16x6` =
    # round 5's tolerance-meeting mode (six bf16 terms forward).  The headline (top-level value) is whatever --dtype / --f32_matmul say:
    # by default the FASTEST mode that meets the tolerances (fp32 storage, three fp16 terms forward, three bf16 terms backward).
    side = {}
    headline = (args.dtype, args.f32_matmul if args.dtype == 'f32' else 'exact')
    if world == 1 and not args.no_f32:
        del step_fn, data, model
        import gc

        def side_run(dtype, matmul, profile):
            gc.collect()
            torch.cuda.empty_cache()
            s2, d2, _, m2 = build_step(args, dtype, None, 1, 0, dev, f32_matmul=matmul)
            w2, k2 = args.warmup, args.steps
            for _ in range(w2):
                f, l = next(d2); s2(f, l)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k2):
                f, l = next(d2); s2(f, l)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            v = global_batch * k2 / e2
            out = dict(value=round(v, 2), unit='images/s', ms_per_step=round(e2 / k2 * 1e3, 2), steps=k2, warmup=w2, dtype=dtype,
                       f32_matmul=matmul if dtype == 'f32' else None, step_mfma_frac=step_mfma_frac(v, dtype, matmul),
                       bn_statistics=('pivoted' if _o.bn_pivot_enabled(torch.float32) else 'raw moments') if dtype == 'f32' else 'raw moments')
            if profile and not args.no_kernel_events and args.prof_steps > 0:
                pr = _o.KernelProfiler()
                _o.PROFILER = pr
                for _ in range(args.prof_steps):
                    f, l = next(d2); s2(f, l)
                torch.cuda.synchronize()
                _o.PROFILER = None
                sm = pr.summary()
                r, fams, other = mode_roofline(sm, args.prof_steps, dtype, matmul)
                out.update(roofline=r, kernels=kernel_table(sm, args.prof_steps))
                if fams is not None:
                    out.update(families=fams)
            del s2, d2, m2
            gc.collect()
            torch.cuda.empty_cache()
            return out

        for key, dtype, matmul, profile in (('speed_mode', 'bf16', 'exact', True), ('f32_mode', 'f32', 'exact', False),
                                            ('parity_mode_bf16x6', 'f32', 'bf16x6_3', False)):
            if (dtype, matmul) != headline:
                side[key] = side_run(dtype, matmul, profile)
        _o.set_f32_matmul('exact')

    parity = None
    if world == 1 and not args.no_parity:
        try:
            del step_fn, data, model
        except NameError:
            pass
        modes = [('value (%s)' % mode_label(*headline), headline[0], headline[1])]
        if not args.no_f32:
            modes += [(k, d, m) for k, d, m in (('speed_mode', 'bf16', 'exact'), ('f32_mode', 'f32', 'exact'), ('parity_mode_bf16x6', 'f32', 'bf16x6_3'))
                      if (d, m) != headline]
        try:
            parity = measured_parity(dev, modes)
        except Exception as e:      # a failed side measurement must be visible in the line, not kill it
            parity = dict(measured_in_run=False, error=repr(e))

    augment = None
    if world == 1:
        # the two-view augmentation on the device (tf2/data_util.py:443-475 x 2 views), NOT part of the timed
        # step (inputs are resident); reported so that the input pipeline's rate can be compared with the step's
        try:
            from simclr_amd import data_util as du
            b_aug, src = args.per_gpu_batch, 256
            raw = torch.randint(0, 256, (b_aug, src, src, 3), dtype=torch.uint8, device=dev)
            prm = torch.from_numpy(du.draw_train_params(b_aug, src, src, args.image_size, args.image_size, 1.0)).to(dev)
            for _ in range(2):
                du.two_view_batch(raw, args.image_size, args.image_size, 1.0, params=prm)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                du.two_view_batch(raw, args.image_size, args.image_size, 1.0, params=prm)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 5
            t0c = time.perf_counter()
            du.draw_train_params(b_aug, src, src, args.image_size, args.image_size, 1.0)
            draw_ms = (time.perf_counter() - t0c) * 1e3
            augment = dict(ms_per_batch=round(ms, 3), images_per_s=round(b_aug / (ms * 1e-3), 1), batch=b_aug,
                           source='uint8 %dx%d' % (src, src), host_param_draw_ms=round(draw_ms, 2),
                           note='crop + bicubic resize + flip + colour jitter + grayscale, both views, on the device; outside the timed step')
            del raw, prm
        except Exception as e:      # never let the side measurement break the benchmark line
            augment = dict(error=repr(e))

    met = None
    if parity and parity.get('modes'):
        met = bool(next(iter(parity['modes'].values())).get('north_star_met'))
    dtype_note = {'bf16': 'bf16 storage and MFMA operands, fp32 accumulation (narrower than the reference: misses north_star)',
                  'f32': 'fp32 storage and accumulation; products: ' + {
                      'exact': 'fp32-input MFMA', 'bf16x3': '3 bf16-piece MFMA terms', 'bf16x6': '6 bf16-piece MFMA terms',
                      'bf16x6_3': '6 bf16-piece terms forward, 3 backward',
                      'f16x3_3': '3 fp16-piece MFMA terms forward (11-bit pieces, ~2^-22 per product), 3 bf16-piece terms backward'}.get(args.f32_matmul, args.f32_matmul)}[args.dtype]
    line = {
        'metric': 'images/sec (whole node), ResNet-%d %dx%s SimCLR pretraining step @%dpx' % (
            args.resnet_depth, args.width_multiplier, '+SK' if args.sk_ratio > 0 else '', args.image_size),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
        'dtype_detail': dtype_note, 'f32_matmul': args.f32_matmul if args.dtype == 'f32' else None,
        # the headline mode's own parity measurement in this run (the `parity` block below): north_star = loss 1e-3 rel, embeddings 1e-5 abs
        'north_star_met': met,
        'data': 'synthetic',
        'config': {'workload': 'ResNet-%d %dx%s, %dx%d, 2 views/image, per-GPU batch %d, global batch %d, '
                               'NT-Xent T=0.1 + linear-eval head + LARS, global BN%s, dp%d'
                               % (args.resnet_depth, args.width_multiplier, '+SK' if args.sk_ratio > 0 else '',
                                  args.image_size, args.image_size, args.per_gpu_batch,
                                  global_batch, ', on-device blur' if args.use_blur else '', world),
                   'global_batch': global_batch, 'parallelism': 'dp%d' % world},
        'step_ms': percentiles(step_ms),
        'step_mfma_frac': step_mfma_frac(value, *headline),
        'flop_per_image': flop_img, 'peak_hbm_gb': round(peak_hbm_gb, 2),
        'per_layer_bound_frac': round(value / (world * 22100.0), 4) if default_cfg and args.dtype == 'bf16' else None,
        'roofline': roofline,
        'families': families,
        'ntxent': ntx,
        'kernels': kernels,
        'speed_mode': side.get('speed_mode'),
        'f32_mode': side.get('f32_mode'),
        'parity_mode_bf16x6': side.get('parity_mode_bf16x6'),
        # measured in this run: every mode's training step against the reference-source fixtures (measured_parity above)
        'parity': parity,
        'allgather': coll,
        'augment': augment,
        'train_metrics': {k: round(v, 5) for k, v in metrics.items()},
    }
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline()
    else:
        line['cpu_baseline'] = None
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    main()
    if dist.is_initialized():
        dist.destroy_process_group()
