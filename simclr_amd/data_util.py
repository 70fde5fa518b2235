"""The on-device part of the augmentation pipeline -- mirror of the pieces of
/root/reference/tf2/data_util.py that `Model.__call__` runs on the accelerator
(tf2/model.py:255-258): `batch_random_blur` -> `random_blur` -> `gaussian_blur`.

The rest of data_util.py (crop/flip/colour jitter inside tf.data) is the host input pipeline and
is out of scope.  Random draws (one sigma ~ U(0.1, 2) per view per batch, a Bernoulli(p) selector
per image) use a torch generator: parity with the reference is distributional for the draws and
exact (tested against the oracle) for the arithmetic given the draws.
"""
import torch

from . import ops

_gen = {}


def _generator(device):
    """One generator per device, seeded with the replica id: every replica draws its own sigma / selectors, like the
    per-replica random ops of the reference (tf2/data_util.py:405, :425-430).  A device generator: the draws never touch
    the host, so the hot path has no host-to-device copy or sync."""
    from .comm import replica_id
    from .resnet import RT
    key = str(device)
    g = _gen.get(key)
    if g is None:
        g = torch.Generator(device=device)
        g.manual_seed(0x51C1 + 7919 * replica_id(RT.strategy))
        _gen[key] = g
    return g


def gaussian_filter(kernel_size, sigma):
    """The 1-D filter of gaussian_blur (tf2/data_util.py:338-343): radius = int(kernel_size / 2),
    size 2*radius+1, exp(-x^2 / (2 sigma^2)) normalised to sum 1 (float32).  `sigma`: a float, or a tensor [k]
    (one filter per row, computed where the tensor lives)."""
    radius = int(kernel_size / 2)
    sig = torch.as_tensor(sigma, dtype=torch.float32)
    x = torch.arange(-radius, radius + 1, dtype=torch.float32, device=sig.device)
    f = torch.exp(-torch.pow(x, 2.0) / (2.0 * torch.pow(sig.reshape(-1, 1), 2.0)))
    f = f / f.sum(dim=1, keepdim=True)
    return f if sig.dim() else f[0]


def batch_random_blur_tensor(images, height, width, blur_probability=0.5, sigmas=None, selectors=None):
    """Fused form used by Model: images float32 [b,H,W,3k] -> blurred/clipped tensor of the same shape.
    sigmas [k] / selectors [k,b] may be given explicitly (tests); otherwise drawn like the reference:
    sigma ~ U(0.1, 2.0) once per view (:405), selector = uniform(0,1) < p per image (:425-430)."""
    b, H, W, C = images.shape
    k = C // 3
    dev = images.device
    if sigmas is None:
        sigmas = 0.1 + 1.9 * torch.rand(k, generator=_generator(dev), device=dev)
    else:
        sigmas = torch.as_tensor(sigmas, dtype=torch.float32).to(dev)
    if selectors is None:
        selectors = (torch.rand(k, b, generator=_generator(dev), device=dev) < blur_probability).float()
    filt = gaussian_filter(height // 10, sigmas)                                                  # :406-407
    sel = torch.as_tensor(selectors, dtype=torch.float32).to(dev).contiguous()
    return ops.batch_blur(images.contiguous(), filt.contiguous(), sel)


def batch_random_blur(images_list, height, width, blur_probability=0.5):
    """Apply efficient batch data transformations (tf2/data_util.py:413-440): list of [b,H,W,3]
    tensors in, list of blurred + clipped tensors out."""
    x = torch.cat(list(images_list), dim=3)
    y = batch_random_blur_tensor(x, height, width, blur_probability)
    return list(torch.split(y, 3, dim=3))
