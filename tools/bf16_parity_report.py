"""bf16 speed mode vs the float64 oracle on one ResNet-50 / 224 px / batch-32 step, for the two kinds of synthetic input
(image-like and the benchmark's i.i.d. noise), with and without fp32 heads.  Writes gpurun_out/bf16_parity.json."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as gc

out = {}
kinds = sys.argv[1:] or ['structured', 'iid']
for inputs in kinds:
    for head in ('same', 'f32'):
        res = gc.check_train_step_fixed(depth=50, image_size=224, batch=32, compute_dtype='bf16', head_dtype=head, inputs=inputs)
        out['%s head=%s' % (inputs, head)] = {r['name'].split(' R50')[0]: r['err'] for r in res}
        for r in res:
            print('%-5s %-70s err=%.3e tol=%.3e' % ('ok' if r['ok'] else 'OVER', r['name'], r['err'], r['tol']), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/bf16_parity.json', 'w'), indent=1)
