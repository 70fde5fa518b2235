#!/usr/bin/env python
"""How many torch-CPU threads should the float64 oracle of the GPU test suite use on the GPU host?  (The suite spends most of its
wall clock in oracle/model_torch.py train_step; with all 128 host threads the small convolutions of the 32 ... 64 px cases
oversubscribe -- bench.py's cpu_baseline found 16 threads 2.7x faster than 128 on the CIFAR-sized model.)

    python tools/oracle_threads_probe.py --out gpurun_out/oracle_threads.json

Times one float64 training step of three representative suite cases at 16 / 32 / 64 / all threads."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

CASES = [('R50 224px b8 (1/4 of the fixed-threshold step)', dict(resnet_depth=50, image_size=224, num_classes=1000), 8),
         ('R50 64px b8', dict(resnet_depth=50, image_size=64, num_classes=1000), 8),
         ('R152 SK 64px b4', dict(resnet_depth=152, image_size=64, num_classes=10, sk_ratio=0.0625), 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/oracle_threads.json')
    ap.add_argument('--threads', default='16,32,64,all')
    a = ap.parse_args()
    from collections import OrderedDict
    from oracle.model_torch import Config, init_model, train_step
    ncpu = os.cpu_count() or 1
    counts = sorted({min(ncpu, ncpu if t == 'all' else int(t)) for t in a.threads.split(',')})
    rows = []
    for name, kw, b in CASES:
        cfg = Config(**kw)
        params, state = init_model(cfg, seed=0)
        p64 = OrderedDict((k, v.double()) for k, v in params.items())
        s64 = OrderedDict((k, v.double()) for k, v in state.items())
        m64 = OrderedDict((k, torch.zeros_like(v)) for k, v in p64.items())
        g = torch.Generator().manual_seed(1)
        images = torch.rand(b, kw['image_size'], kw['image_size'], 6, generator=g).double()
        labels = torch.nn.functional.one_hot(torch.randint(0, kw['num_classes'], (b,), generator=g), kw['num_classes']).double()
        for n in counts:
            torch.set_num_threads(n)
            t0 = time.time()
            train_step(cfg, p64, s64, m64, images, labels, 0.1)
            rows.append(dict(case=name, threads=n, seconds=round(time.time() - t0, 2)))
            print(rows[-1], flush=True)
    json.dump(dict(host_cpus=ncpu, rows=rows), open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
