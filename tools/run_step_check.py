"""End-to-end step parity on the GPU box: python tools/run_step_check.py"""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gpu_checks as gc

results = []
def run(**kw):
    t = time.time()
    try:
        for d in gc.check_train_step(**kw):
            d['sec'] = round(time.time() - t, 1)
            results.append(d)
            print('%-4s %-62s err=%.3e tol=%.3e ref_err=%.3e %s' % ('ok' if d['ok'] else 'FAIL', d['name'], d['err'], d['tol'], d.get('scale', 0),
                  {k: v for k, v in d.items() if k in ('worst', 'missing', 'extra', 'value', 'ref')}), flush=True)
    except Exception as e:
        traceback.print_exc()
        print('EXC', kw, repr(e), flush=True)
for r in gc.check_probes():
    print('ok' if r['ok'] else 'FAIL', r['name'], r['err'])
print('ds_read_tr16 map (lane -> 4 elems), lanes 0..19 and 60..63:')
m = gc.probe_ds_read_tr16()
print(m[:20].tolist(), m[60:].tolist())
os.makedirs('gpurun_out', exist_ok=True)
json.dump(m.tolist(), open('gpurun_out/ds_read_tr16_map.json', 'w'))
for r in gc.check_avgpool2(2, 8, 64, 2, torch.float32) + gc.check_avgpool2(2, 7, 64, 2, torch.float32) + gc.check_avgpool2(2, 7, 64, 1, torch.bfloat16) + gc.check_avgpool2(3, 6, 128, 1, torch.float32):
    print('ok' if r['ok'] else 'FAIL', r['name'], r['err'], r['tol'])
run(depth=50, image_size=64, batch=4, compute_dtype='f32', num_classes=1000, randomize_bn=False, sk_ratio=0.0625)
run(depth=50, image_size=64, batch=4, compute_dtype='f32', num_classes=1000, randomize_bn=True, sk_ratio=0.0625, width_multiplier=2)
run(depth=50, image_size=64, batch=4, compute_dtype='bf16', num_classes=1000, randomize_bn=False, sk_ratio=0.0625)
run(depth=18, image_size=64, batch=8, compute_dtype='f32', sk_ratio=0.0625)
run(depth=18, image_size=32, batch=16, compute_dtype='f32', steps=2)
run(depth=18, image_size=32, batch=16, compute_dtype='bf16')
run(depth=18, image_size=32, batch=16, compute_dtype='bf16', randomize_bn=False)
run(depth=50, image_size=64, batch=4, compute_dtype='f32', num_classes=1000)
run(depth=50, image_size=64, batch=4, compute_dtype='f32', num_classes=1000, randomize_bn=False)
run(depth=50, image_size=64, batch=4, compute_dtype='bf16', num_classes=1000, randomize_bn=False)
json.dump(results, open('gpurun_out/step_checks.json', 'w'), indent=1, default=str)
