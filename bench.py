#!/usr/bin/env python
"""Benchmark of the SimCLR pretraining step on MI355X (BASELINE.json metric: images/sec).

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one full pretraining step (tf2/run.py:557-622): two-view ResNet-50 1x forward +
backward at 224 px, projection head, NT-Xent, linear-eval head, LARS -- on a synthetic batch that
is resident in HBM before the timed region.  One "image" = one dataset image = two views
(BASELINE.md).  Per-GPU batch is fixed at 512 images (BASELINE.json configs[1] at N=1,
configs[2] = global 4096 at N=8) => weak scaling; `value` = global_batch * K / max-over-ranks time.

The JSON line also carries:
  roofline      -- the dominant kernel family measured LIVE with HIP events on the launch stream
                   inside the timed region: algorithmic FLOPs (2*M*N*K per launch) and
                   algorithmic bytes (operands read once, output written once) / summed launch time.
                   Its arithmetic intensity (~160 FLOP/B for ResNet-50 1x) is below the bf16 ridge
                   (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B), so the binding roof is HBM; the MFMA fraction
                   is reported next to it.  `traffic` = PMC bytes per launch (profiles/).
  cpu_baseline  -- the CPU oracle (torch-CPU restatement of the TF2 reference; TensorFlow is not
                   installed) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
FLOP_PER_IMAGE = 49.15e9    # 2 views x (fwd+dgrad+wgrad), encoder + head


def cpu_baseline(seconds_budget=25.0):
    """Time the oracle's full training step (same R50/224 step, small batch) on the host cores."""
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, train_step
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    cfg = Config(resnet_depth=50, image_size=224, num_classes=1000)
    params, state = init_model(cfg, seed=2)
    momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
    b = 8
    images = torch.rand(b, 224, 224, 6)
    labels = torch.nn.functional.one_hot(torch.randint(0, 1000, (b,)), 1000).float()
    train_step(cfg, params, state, momenta, images, labels, 0.1)       # warm-up (allocator, MKL init)
    n, t0 = 0, time.time()
    while True:
        train_step(cfg, params, state, momenta, images, labels, 0.1)
        n += 1
        if time.time() - t0 > seconds_budget * 0.6 or n >= 4:
            break
    dt = time.time() - t0
    return dict(value=round(b * n / dt, 3), unit='images/s', cores=cores, kind='port',
                sample='%d full train steps of the same ResNet-50 1x @224 step at batch %d '
                       '(torch-CPU fp32 restatement of tf2/, %.1f s)' % (n, b, dt))


PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def make_roofline(kernel, flops, nbytes, total_ms, launches, steps, mfma_peak_tflops, traffic, traffic_src):
    """Roofline entry of the dominant kernel family.  `flops` / `nbytes`: algorithmic totals over `launches`
    launches that took `total_ms` (HIP events on the launch stream).  The binding roof is decided the usual way:
    arithmetic intensity below the ridge (MFMA peak / HBM peak) => 'hbm', else 'mfma'; `achieved` / `peak` / `frac`
    are quoted for that roof, the other one is kept under explicit names."""
    sec = max(total_ms, 1e-9) * 1e-3
    tflops = flops / sec / 1e12
    gbps = nbytes / sec / 1e9
    intensity = flops / max(nbytes, 1.0)
    ridge = mfma_peak_tflops * 1e12 / (PEAK_HBM_GBPS * 1e9)
    hbm = intensity < ridge
    d = dict(bound='hbm' if hbm else 'mfma', kernel=kernel)
    if hbm:
        d.update(achieved=round(gbps, 1), peak=PEAK_HBM_GBPS, unit='GB/s', frac=round(gbps / PEAK_HBM_GBPS, 4))
    else:
        d.update(achieved=round(tflops, 2), peak=mfma_peak_tflops, unit='TFLOP/s', frac=round(tflops / mfma_peak_tflops, 4))
    d.update(traffic=traffic, traffic_source=traffic_src,
             algorithmic_bytes_per_launch=round(nbytes / max(launches, 1)),
             flops_per_launch_avg=flops / max(launches, 1),
             arithmetic_intensity_flop_per_byte=round(intensity, 1), ridge_flop_per_byte=round(ridge, 1),
             achieved_tflops=round(tflops, 2), mfma_frac=round(tflops / mfma_peak_tflops, 4),
             achieved_alg_gbps=round(gbps, 1), hbm_frac=round(gbps / PEAK_HBM_GBPS, 4),
             avg_launch_us=round(total_ms * 1e3 / max(launches, 1), 2), launches_per_step=launches // max(steps, 1),
             ms_per_step=round(total_ms / max(steps, 1), 3))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--per_gpu_batch', type=int, default=512)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--resnet_depth', type=int, default=50)
    ap.add_argument('--image_size', type=int, default=224)
    ap.add_argument('--width_multiplier', type=int, default=1)
    ap.add_argument('--sk_ratio', type=float, default=0.0)
    ap.add_argument('--use_blur', action='store_true', help='include the on-device batch_random_blur (reference default)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_kernel_events', action='store_true')
    args = ap.parse_args()

    from simclr_amd import model as model_lib
    from simclr_amd import ops
    from simclr_amd.flags import FLAGS
    from simclr_amd.resnet import RT
    from simclr_amd.run import init_distributed, make_single_step, synthetic_batches

    strategy = init_distributed()
    world = 1 if strategy is None else strategy.num_replicas_in_sync
    rank = 0 if strategy is None else strategy.rank
    assert world == args.gpus, 'launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    dev = torch.device('cuda', torch.cuda.current_device())

    global_batch = args.per_gpu_batch * world
    FLAGS.reset()
    FLAGS.update(resnet_depth=args.resnet_depth, width_multiplier=args.width_multiplier, image_size=args.image_size,
                 sk_ratio=args.sk_ratio, train_batch_size=global_batch, compute_dtype=args.dtype, use_blur=args.use_blur,
                 learning_rate=0.075, learning_rate_scaling='sqrt', weight_decay=1e-6,
                 temperature=0.1, hidden_norm=True, global_bn=True, lineareval_while_pretraining=True)
    RT.reset()
    RT.strategy = strategy
    RT.device = dev
    num_classes = 1000
    model = model_lib.Model(num_classes)
    schedule = model_lib.WarmUpAndCosineDecay(FLAGS.learning_rate, 1281167)
    optimizer = model_lib.build_optimizer(schedule)
    optimizer.iterations = 1000   # past step 0 so the warm-up LR is non-zero (weights really move)
    step_fn = make_single_step(model, optimizer, strategy)
    data = synthetic_batches(args.per_gpu_batch, args.image_size, num_classes, dev, seed=rank)

    def sync():
        if strategy is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        f, l = next(data)
        step_fn(f, l)
    sync()
    prof = None
    if not args.no_kernel_events:
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f, l = next(data)
        step_fn(f, l)
    sync()
    elapsed = time.perf_counter() - t0
    ops.PROFILER = None
    if strategy is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = global_batch * args.steps / elapsed
    metrics = {k: v.result() for k, v in step_fn.metrics.items()}

    if rank != 0:
        return
    peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
    roofline = None
    kernels = {}
    if prof is not None:
        summ = prof.summary()
        for fam, d in summ.items():
            kernels[fam] = dict(launches_per_step=d['launches'] // args.steps,
                                ms_per_step=round(d['ms'] / args.steps, 3),
                                tflops=round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2) if d['ms'] > 0 else None,
                                alg_gbps=round(d['bytes'] / (d['ms'] * 1e-3) / 1e9, 1) if d['ms'] > 0 else None)
        # dominant family = most GPU time; the fwd and dgrad launches are the same kernel template
        fams = {'conv_igemm': [k for k in summ if k.startswith('conv_igemm')], 'conv_wgrad': ['conv_wgrad']}
        best, best_ms = None, -1.0
        for name, members in fams.items():
            ms = sum(summ[m]['ms'] for m in members if m in summ)
            if ms > best_ms:
                best, best_ms = name, ms
        members = [m for m in fams[best] if m in summ]
        fl = sum(summ[m]['flops'] for m in members)
        nl = sum(summ[m]['launches'] for m in members)
        achieved = fl / (best_ms * 1e-3) / 1e12 if best_ms > 0 else 0.0
        # HBM traffic per launch of the same kernel family: from the committed rocprofv3 --pmc passes
        # (FETCH_SIZE / WRITE_SIZE collected separately, FETCH x2 for wide loads -- profiles/r01_pmc_traffic.json);
        # PMC counters cannot be sampled from inside this process, so this is null when the file is absent
        # or describes another kernel family / configuration.
        traffic, traffic_src = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
        default_cfg = (args.resnet_depth == 50 and args.image_size == 224 and args.width_multiplier == 1 and
                       args.sk_ratio == 0 and args.per_gpu_batch == 512 and args.dtype == 'bf16')
        if best == 'conv_igemm' and default_cfg and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = round(tj['traffic_bytes_per_launch'])
                traffic_src = 'profiles/r01_pmc_traffic.json'
            except Exception:
                traffic = None
        by = sum(summ[m]['bytes'] for m in members)
        roofline = make_roofline(best, fl, by, best_ms, nl, args.steps, peak, traffic, traffic_src)
    line = {
        'metric': 'images/sec (whole node), ResNet-%d %dx%s SimCLR pretraining step @%dpx' % (
            args.resnet_depth, args.width_multiplier, '+SK' if args.sk_ratio > 0 else '', args.image_size),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': 'ResNet-%d %dx%s, %dx%d, 2 views/image, per-GPU batch %d, global batch %d, '
                               'NT-Xent T=0.1 + linear-eval head + LARS, global BN%s, dp%d'
                               % (args.resnet_depth, args.width_multiplier, '+SK' if args.sk_ratio > 0 else '',
                                  args.image_size, args.image_size, args.per_gpu_batch,
                                  global_batch, ', on-device blur' if args.use_blur else '', world),
                   'global_batch': global_batch, 'parallelism': 'dp%d' % world},
        'step_mfma_frac': round(value * FLOP_PER_IMAGE / (world * peak * 1e12), 4)
        if args.resnet_depth == 50 and args.image_size == 224 and args.width_multiplier == 1 and args.sk_ratio == 0 else None,
        'roofline': roofline,
        'kernels': kernels,
        'train_metrics': {k: round(v, 5) for k, v in metrics.items()},
    }
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline()
    else:
        line['cpu_baseline'] = None
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
    if dist.is_initialized():
        dist.destroy_process_group()
