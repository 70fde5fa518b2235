"""Per-tensor gradient errors of one training step vs the float64 oracle (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from oracle.model_torch import Config, init_model, train_step
from simclr_amd import model as model_lib
from simclr_amd.flags import FLAGS
from simclr_amd.resnet import RT
from simclr_amd.run import make_single_step

depth, image_size, batch, nc, sk, w = 18, 64, 8, 10, 0.0625, 1
if len(sys.argv) > 1:
    depth, image_size, batch, sk, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
cfg = Config(resnet_depth=depth, image_size=image_size, num_classes=nc, weight_decay=1e-4, sk_ratio=sk, width_multiplier=w)
params, state = init_model(cfg, seed=0, randomize_bn=True)
g = torch.Generator().manual_seed(1)
FLAGS.reset(); FLAGS.update(resnet_depth=depth, image_size=image_size, compute_dtype='f32', use_blur=False, weight_decay=1e-4,
                            train_batch_size=batch, sk_ratio=sk, width_multiplier=w)
RT.reset(); RT.device = torch.device('cuda')
model = model_lib.Model(nc)
with torch.no_grad():
    model(torch.zeros(2, image_size, image_size, 6, device='cuda'), training=True)
allv = dict(params); allv.update(state)
for v in model.variables:
    v.value.copy_(allv[v.name].cuda())
RT.weights_version += 1
opt = model_lib.build_optimizer(0.1)
step = make_single_step(model, opt, None)
images = torch.rand(batch, image_size, image_size, 6, generator=g)
labels = torch.nn.functional.one_hot(torch.randint(0, nc, (batch,), generator=g), nc).float()
p64 = OrderedDict((k, v.double()) for k, v in params.items()); s64 = OrderedDict((k, v.double()) for k, v in state.items())
m64 = OrderedDict((k, torch.zeros_like(v)) for k, v in p64.items())
_, _, _, t64 = train_step(cfg, p64, s64, m64, images.double(), labels.double(), 0.1)
m32 = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
_, _, _, t32 = train_step(cfg, params, state, m32, images, labels, 0.1)
out = step(images.cuda(), {'labels': labels.cuda()})
torch.cuda.synchronize()
print('loss', float(out['con_loss'].value), float(t64['con_loss']))
byname = {v.name: v for v in model._flat_order}
rows = []
for i, (k, ref) in enumerate(t64['grads'].items()):
    if ref is None or float(ref.abs().max()) < 1e-12: continue
    e = float((byname[k].grad.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    e32 = float((t32['grads'][k].double() - ref).abs().max()) / float(ref.abs().max())
    rows.append((i, e, e32, k, tuple(ref.shape)))
for r in rows:
    flag = '***' if r[1] > 20 * r[2] + 1e-4 else ''
    print('%3d mine=%.2e oracle32=%.2e %s %s %s' % (r[0], r[1], r[2], r[3].replace('model/resnet/', ''), r[4], flag))
