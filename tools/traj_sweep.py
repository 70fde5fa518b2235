"""bf16-vs-f32 training trajectory (tests/gpu_checks.py check_bf16_trajectory) under a few (lr, pool, window) settings:
prints the worst window deviations, so the test's setting can be chosen with the data in hand."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as gc

out = {}
for (lr, pool, window, after) in [(0.3, 8, 10, 20), (0.3, 8, 16, 20), (0.1, 16, 16, 20), (0.1, 32, 16, 20), (0.05, 32, 16, 20)]:
    res = gc.check_bf16_trajectory(lr=lr, pool=pool, window=window, after=after)
    row = {r['name'].split(' R18')[0]: r['err'] for r in res}
    extra = [r for r in res if 'f32_input_rounding_loss_rel' in r][0]
    row['f32_input_rounding_loss_rel'] = extra['f32_input_rounding_loss_rel']
    row['f32_input_rounding_acc_abs'] = extra['f32_input_rounding_acc_abs']
    row['f32_last'] = extra['f32_last']
    out['lr%g pool%d win%d' % (lr, pool, window)] = row
    print('lr %.2f pool %2d window %2d: loss %.3f -> %.3f | bf16 worst window loss rel %.4f acc abs %.4f | f32-on-rounded-inputs %.4f / %.4f' % (
        lr, pool, window, [r for r in res if r['name'].startswith('traj_loss_falls')][0]['first'], extra['f32_last'],
        row['traj_contrast_loss_window_rel bf16 vs f32'], row['traj_contrast_acc_window_abs bf16 vs f32'],
        row['f32_input_rounding_loss_rel'], row['f32_input_rounding_acc_abs']), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/traj_sweep.json', 'w'), indent=1)
