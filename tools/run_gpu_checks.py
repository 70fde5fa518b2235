"""First-contact GPU diagnostics: run every kernel parity check, print all errors, dump JSON.

Usage (on the GPU box): python tools/run_gpu_checks.py [--quick]
Writes gpurun_out/gpu_checks.json.  Exit code 0 even on failures (this is a report, the
gate is `pytest -m gpu`).
"""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests import gpu_checks as gc  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def main():
    results = []

    def run(fn, *a, **k):
        t = time.time()
        try:
            r = fn(*a, **k)
            torch.cuda.synchronize()
            for d in r:
                d['sec'] = round(time.time() - t, 2)
                results.append(d)
                print('%-4s %-70s err=%.3e tol=%.3e scale=%.3e nbad=%d/%d' % (
                    'ok' if d['ok'] else 'FAIL', d['name'], d['err'], d['tol'], d['scale'], d['nbad'], d['numel']),
                    flush=True)
        except Exception as e:  # noqa
            traceback.print_exc()
            results.append(dict(name='%s%r' % (fn.__name__, a), ok=False, err=-1, exc=repr(e)))
            print('EXC  %s%r: %r' % (fn.__name__, a, e), flush=True)

    print(torch.cuda.get_device_name(0), flush=True)
    run(gc.check_probes)
    try:
        m = gc.probe_ds_read_tr16()
        print('ds_read_tr16 lanes 0..19:\n', m[:20].tolist(), flush=True)
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(m.tolist(), open('gpurun_out/ds_read_tr16_map.json', 'w'))
    except Exception:
        traceback.print_exc()
    run(gc.check_ntxent_closed_forms)
    for n, R, D, rank in [(64, 1, 128, 0), (96, 1, 64, 0), (32, 4, 128, 2), (512, 1, 128, 0), (64, 2, 256, 1)]:
        run(gc.check_ntxent, n, R, D=D, rank=rank)
    run(gc.check_ntxent, 64, 1, hidden_norm=False, temperature=1.0)
    run(gc.check_lars)
    run(gc.check_lars, classic=False, nesterov=True)
    run(gc.check_lars, classic=True, nesterov=True)
    for dt in (F32, BF):
        for (V, H, Cin, Cout, k, s) in [(2, 8, 64, 64, 1, 1), (3, 14, 64, 128, 3, 1), (2, 15, 128, 64, 3, 2),
                                        (2, 16, 64, 256, 1, 2), (3, 9, 128, 192, 3, 1), (130, 1, 128, 64, 1, 1),
                                        (2, 12, 256, 128, 3, 2)]:
            run(gc.check_conv, V, H, H, Cin, Cout, k, s, dt)
        for (V, H, Cin, Cout, k, mode, acc) in [(3, 9, 128, 64, 1, 2, 0), (2, 14, 64, 128, 3, 2, 0), (3, 8, 256, 64, 1, 1, 1), (2, 7, 64, 64, 3, 1, 0)]:
            run(gc.check_dgrad_bn, V, H, Cin, Cout, k, dt, mode, acc)
        run(gc.check_stem, 4, 32, 7, 2, 64, dt)
        run(gc.check_stem, 4, 16, 3, 1, 64, dt)
        run(gc.check_stem, 2, 224, 7, 2, 64, dt)
        for relu, resid in [(True, None), (False, None), (True, 'identity'), (True, 'bn')]:
            run(gc.check_bn, (6, 7, 5), 64, dt, relu, resid)
        run(gc.check_bn, (37,), 2048, dt, False, None)
        run(gc.check_bn, (500,), 128, dt, True, None)
        run(gc.check_pool, 2, 16, 64, dt)
        run(gc.check_pool, 2, 15, 64, dt)
        run(gc.check_sup_head, 64, 1000, 1008, dt)
        run(gc.check_sup_head, 16, 10, 16, dt)
    nfail = sum(1 for r in results if not r['ok'])
    print('TOTAL %d checks, %d failed' % (len(results), nfail), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(results, open('gpurun_out/gpu_checks.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
