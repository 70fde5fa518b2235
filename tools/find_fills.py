"""Debug aid: which Python call sites launch torch's OWN device kernels (fills, copies, elementwise, reductions) inside a
steady-state training step -- everything that is not a libsimclr_hip.so launch.
python tools/find_fills.py  -> aten ops with device time, grouped by the top Python frames (ResNet-50, 64 px, batch 16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from simclr_amd import model as model_lib  # noqa: E402
from simclr_amd.flags import FLAGS  # noqa: E402
from simclr_amd.resnet import RT  # noqa: E402
from simclr_amd.run import make_single_step  # noqa: E402

dev = torch.device('cuda', 0)
FLAGS.reset()
FLAGS.update(resnet_depth=50, image_size=64, train_batch_size=16, compute_dtype=(sys.argv[1] if len(sys.argv) > 1 else 'bf16'), use_blur=False)
RT.reset(); RT.device = dev
model = model_lib.Model(10)
opt = model_lib.build_optimizer(0.1)
step = make_single_step(model, opt, None)
x = torch.rand(16, 64, 64, 6, device=dev)
lab = torch.nn.functional.one_hot(torch.randint(0, 10, (16,), device=dev), 10).float()
for i in range(3):
    step(x, {'labels': lab})
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, {'labels': lab})
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    dt = getattr(e, 'device_time_total', 0) or getattr(e, 'cuda_time_total', 0)
    if not e.key.startswith('aten::') or dt <= 0:
        continue
    own = [s for s in e.stack if 'simclr_amd' in s or 'bench.py' in s][:3]
    rows.append((e.count, dt, e.key, ' <- '.join(s.split('/')[-1] for s in own)))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print('aten ops with device kernels in one step: %d' % tot)
for c, dt, k, st in rows[:60]:
    print('%4d x %-28s %8.1f us  %s' % (c, k, dt, st))
