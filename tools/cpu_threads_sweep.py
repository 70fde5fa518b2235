#!/usr/bin/env python
"""One-off thread sweep of bench.py's cpu_baseline: BASELINE configs[0] (ResNet-18, CIFAR 32x32, batch 256, one replica,
fp32; the torch-CPU restatement of tf2/run.py:557-622) at 16 / 32 / 64 / all host threads -- 1 warm-up + 3 timed steps each --
so that the thread count bench.py uses is evidence, not a comment.

    python tools/cpu_threads_sweep.py --out gpurun_out/cpu_threads.json        (copy into profiles/rNN_cpu_threads.json)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/cpu_threads.json')
    ap.add_argument('--threads', default='16,32,64,all')
    ap.add_argument('--steps', type=int, default=3)
    a = ap.parse_args()
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, train_step
    ncpu = os.cpu_count() or 1
    counts = sorted({min(ncpu, ncpu if t == 'all' else int(t)) for t in a.threads.split(',')})
    cfg = Config(resnet_depth=18, image_size=32, num_classes=10)
    torch.manual_seed(0)
    images = torch.rand(256, 32, 32, 6)
    labels = torch.nn.functional.one_hot(torch.randint(0, 10, (256,)), 10).float()
    rows = []
    for n in counts:
        torch.set_num_threads(n)
        params, state = init_model(cfg, seed=2)
        momenta = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        train_step(cfg, params, state, momenta, images, labels, 0.1)             # warm-up
        t0 = time.time()
        for _ in range(a.steps):
            train_step(cfg, params, state, momenta, images, labels, 0.1)
        dt = (time.time() - t0) / a.steps
        rows.append(dict(threads=n, s_per_step=round(dt, 3), images_per_s=round(256 / dt, 2)))
        print(rows[-1], flush=True)
    best = max(rows, key=lambda r: r['images_per_s'])
    json.dump(dict(host_cpus=ncpu, config='BASELINE configs[0]: ResNet-18, CIFAR 32x32, batch 256, 1 replica, fp32 (oracle/model_torch.py)',
                   steps_timed=a.steps, warmup=1, rows=rows, best_threads=best['threads']), open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
