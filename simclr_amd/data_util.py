"""The augmentation pipeline on the device -- mirror of /root/reference/tf2/data_util.py.

* `batch_random_blur` -> `random_blur` -> `gaussian_blur`: what `Model.__call__` runs on the accelerator
  (tf2/model.py:255-258, tf2/data_util.py:323-440).
* `preprocess_for_train_batch` / `two_view_batch` / `preprocess_for_eval_batch`: the per-image part of the input
  pipeline (random-resized-crop with bicubic resize, flip, colour jitter in random order, random grayscale;
  tf2/data_util.py:53-320, :362-390, :443-499 and the two-views concat of tf2/data.py:52-62) for a whole batch of
  decoded images at once -- at >6 000 images/s per GPU a host tf.data-style pipeline is the limiter.
  The tfds reader / JPEG decode stay outside (out of scope).

Random draws (crop boxes, coins, op order, factors; sigma and selectors of the blur) come from numpy / torch
generators: parity with the reference is distributional for the draws and exact (tested against the oracle,
oracle/augment.py and oracle/blur.py) for the arithmetic given the draws.
"""
import numpy as np
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


# --------------------------------------------------------------------------- crop / flip / colour jitter (input pipeline)
CROP_PROPORTION = 0.875            # tf2/data_util.py:22
PARAM_FIELDS = ('crop_y', 'crop_x', 'crop_h', 'crop_w', 'flip', 'jitter_on', 'perm0', 'perm1', 'perm2', 'perm3',
                'brightness', 'contrast', 'saturation', 'hue', 'gray_on', 'pad')
_np_rng = {}


def _rng():
    """numpy generator of this replica (seeded with the replica id, like the per-replica tf.data pipelines)."""
    from .comm import replica_id
    from .resnet import RT
    r = replica_id(RT.strategy)
    if r not in _np_rng:
        _np_rng[r] = np.random.default_rng(0xA06 + 104729 * r)
    return _np_rng[r]


def _lrint(x):
    return np.rint(x).astype(np.int64)


def sample_crop_boxes(rng, heights, widths, out_h, out_w, max_attempts=100):
    """tf.image.sample_distorted_bounding_box as crop_and_resize calls it (tf2/data_util.py:298-320: whole-image box,
    min_object_covered 0.1, aspect ratio in [3/4, 4/3] * out_w/out_h, area in [0.08, 1], 100 attempts, then the whole
    image), vectorised over the images; the per-attempt arithmetic follows TensorFlow's GenerateRandomCrop
    (float32 areas, lrintf roundings).  heights / widths: int arrays [m].  Returns int64 [m, 4] = (y, x, h, w)."""
    Hh = np.asarray(heights, dtype=np.int64)
    Ww = np.asarray(widths, dtype=np.int64)
    m = Hh.shape[0]
    f32 = np.float32
    ar0 = out_w / out_h
    lo, hi = f32(3. / 4 * ar0), f32(4. / 3. * ar0)
    box = np.stack([np.zeros(m, np.int64), np.zeros(m, np.int64), Hh, Ww], 1)
    todo = np.ones(m, bool)
    min_area = f32(0.08) * Ww.astype(f32) * Hh.astype(f32)
    max_area = f32(1.0) * Ww.astype(f32) * Hh.astype(f32)
    for _ in range(max_attempts):
        if not todo.any():
            break
        ar = (rng.random(m).astype(f32) * (hi - lo) + lo).astype(np.float64)
        h = _lrint(np.sqrt(min_area / ar.astype(f32)))
        max_h = _lrint(np.sqrt(max_area / ar.astype(f32)))
        over = _lrint(max_h * ar) > Ww
        alt = ((Ww + 0.5 - 0.0000001) / ar).astype(np.int64)
        alt = np.where(_lrint(alt * ar) > Ww, alt - 1, alt)
        max_h = np.where(over, alt, max_h)
        max_h = np.minimum(max_h, Hh)
        h = np.minimum(h, max_h)
        h = h + np.floor(rng.random(m) * (max_h - h + 1)).astype(np.int64) * (h < max_h)
        w = _lrint(h * ar)
        area = (w * h).astype(np.float64)
        small = area < min_area
        h = np.where(small, h + 1, h); w = np.where(small, _lrint(h * ar), w); area = (w * h).astype(np.float64)
        big = area > max_area
        h = np.where(big, h - 1, h); w = np.where(big, _lrint(h * ar), w); area = (w * h).astype(np.float64)
        ok = ~((area < min_area) | (area > max_area) | (w > Ww) | (h > Hh) | (w <= 0) | (h <= 0))
        y = np.floor(rng.random(m) * np.maximum(Hh - h, 1)).astype(np.int64) * (h < Hh)
        x = np.floor(rng.random(m) * np.maximum(Ww - w, 1)).astype(np.int64) * (w < Ww)
        ok &= (w * h) / (Ww * Hh).astype(np.float64) >= 0.1            # min_object_covered of the whole-image box
        take = todo & ok
        box[take] = np.stack([y, x, h, w], 1)[take]
        todo &= ~ok
    return box


def draw_train_params(b, heights, widths, height, width, color_jitter_strength=1.0, crop=True, flip=True, views=2,
                      rng=None):
    """Random draws of preprocess_for_train (tf2/data_util.py:443-475) for `views` views of each of `b` images.
    heights / widths: per-image source sizes (int or arrays).  Returns float32 [b, views, 16] laid out as PARAM_FIELDS."""
    rng = _rng() if rng is None else rng
    m = b * views
    Hh = np.repeat(np.broadcast_to(np.asarray(heights, np.int64), (b,)), views)
    Ww = np.repeat(np.broadcast_to(np.asarray(widths, np.int64), (b,)), views)
    p = np.zeros((m, len(PARAM_FIELDS)), np.float32)
    p[:, 0:4] = sample_crop_boxes(rng, Hh, Ww, height, width) if crop else np.stack([0 * Hh, 0 * Ww, Hh, Ww], 1)
    if flip:
        p[:, 4] = rng.random(m) < 0.5                                              # :463
    s = float(color_jitter_strength)
    if s > 0:                                                                      # :380-389
        p[:, 5] = rng.random(m) < 0.8
        p[:, 6:10] = np.argsort(rng.random((m, 4)), axis=1)                        # a uniform random order, :168
        bb, cc, ss, hh = 0.8 * s, 0.8 * s, 0.8 * s, 0.2 * s                        # :70-73
        p[:, 10] = rng.uniform(max(1.0 - bb, 0.0), 1.0 + bb, m)                    # :37-39
        p[:, 11] = rng.uniform(1 - cc, 1 + cc, m)
        p[:, 12] = rng.uniform(1 - ss, 1 + ss, m)
        p[:, 13] = rng.uniform(-hh, hh, m)
        p[:, 14] = rng.random(m) < 0.2
    return p.reshape(b, views, len(PARAM_FIELDS))


def _sizes(images, sizes):
    b, Hs, Ws, _ = images.shape
    if sizes is None:
        return np.full(b, Hs), np.full(b, Ws)
    sz = np.asarray(sizes.cpu() if torch.is_tensor(sizes) else sizes)
    return sz[:, 0], sz[:, 1]


def preprocess_for_train_batch(images, height, width, color_jitter_strength=0., crop=True, flip=True, impl='simclrv2',
                               sizes=None, views=1, params=None):
    """Batched preprocess_for_train (tf2/data_util.py:443-475).  images: device tensor [b, Hs, Ws, 3], uint8 or float32
    in [0,1] -- a canvas; `sizes` [b, 2] gives each image's valid (height, width) inside it (default: the whole canvas).
    Returns float32 [b, height, width, 3*views]."""
    if impl != 'simclrv2':
        raise ValueError('Unknown impl {} for random brightness.'.format(impl))      # the build ships simclrv2 only
    b = images.shape[0]
    if params is None:
        hs, ws = _sizes(images, sizes)
        params = draw_train_params(b, hs, ws, height, width, color_jitter_strength, crop, flip, views)
    if not torch.is_tensor(params):
        params = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32))
    params = params.to(images.device, non_blocking=True)
    return ops.augment_views(images.contiguous(), params.contiguous(), height, width)


def two_view_batch(images, height, width, color_jitter_strength=1.0, sizes=None, params=None):
    """tf2/data.py:52-62: `xs = [preprocess(image), preprocess(image)]; concat(xs, -1)` for the whole batch ->
    [b, height, width, 6], the tensor Model.__call__ consumes."""
    return preprocess_for_train_batch(images, height, width, color_jitter_strength, sizes=sizes, views=2, params=params)


def center_crop_boxes(heights, widths, height, width, crop_proportion=CROP_PROPORTION):
    """tf2/data_util.py:175-243 (_compute_crop_shape + center_crop offsets) for arrays of image sizes."""
    f32 = np.float32
    ih, iw = np.asarray(heights).astype(f32), np.asarray(widths).astype(f32)
    ar = width / height
    wider = ar > iw / ih
    ch = np.where(wider, np.rint(f32(crop_proportion / ar) * iw), np.rint(f32(crop_proportion) * ih)).astype(np.int64)
    cw = np.where(wider, np.rint(f32(crop_proportion) * iw), np.rint(f32(crop_proportion * ar) * ih)).astype(np.int64)
    Hh, Ww = np.asarray(heights, np.int64), np.asarray(widths, np.int64)
    return np.stack([((Hh - ch) + 1) // 2, ((Ww - cw) + 1) // 2, ch, cw], 1)


def preprocess_for_eval_batch(images, height, width, crop=True, sizes=None):
    """Batched preprocess_for_eval (tf2/data_util.py:478-499): central crop (CROP_PROPORTION) + bicubic resize + clip."""
    b = images.shape[0]
    hs, ws = _sizes(images, sizes)
    p = np.zeros((b, 1, len(PARAM_FIELDS)), np.float32)
    p[:, 0, 0:4] = center_crop_boxes(hs, ws, height, width) if crop else np.stack([0 * hs, 0 * ws, hs, ws], 1)
    return preprocess_for_train_batch(images, height, width, params=p)
