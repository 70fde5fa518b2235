"""Seeded inputs of the "well-conditioned" reference fixtures (`*_img` cases of make_reference_golden.py).  Pure numpy: no import of
the oracle, of the product or of the reference, so that every party -- the reference's source on oracle/tfshim.py, the oracle, the
device tests and bench.py's in-run parity block -- can derive the SAME variables and images from names and shapes alone.

Variables are keyed by their reference NAME (tf2/resnet.py / tf2/model.py layer construction, e.g.
'resnet/block_group1/bottleneck_block/conv2d_fixed_padding_1/conv2d_1/kernel:0'): the value does not depend on creation order.

  * kernels (random in the reference, tf2/resnet.py:201 VarianceScaling / tf2/model.py:145 RandomNormal(stddev=.01)): drawn here
    from those distributions (fan-in truncated normal / N(0, 0.01^2));
  * every other variable keeps the INITIAL value its owner gave it (gamma 1 or 0 on block tails, beta / bias 0, moving mean 0,
    moving variance 1; tf2/resnet.py:42-48) and -- with perturb=True -- gets a seeded offset so that no branch hides behind zeros.
"""
import math
import zlib

import numpy as np


def _rng(name, salt):
    return np.random.default_rng([zlib.crc32(name.encode()), salt])


def variable_value(name, init, perturb):
    """float64 value of the variable called `name` (reference name without the 'model/' scope); `init`: its initial value."""
    init = np.asarray(init, dtype=np.float64)
    rng = _rng(name, 2024)
    if name.endswith('kernel:0'):
        if init.ndim == 4:
            kh, kw, cin, _ = init.shape
            std = math.sqrt(1.0 / (kh * kw * cin)) / .87962566103423978
            v = rng.standard_normal(init.shape)
            bad = np.abs(v) > 2
            while bad.any():                                   # truncated normal: resample outside two sigma
                v[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2
            return v * std
        return rng.standard_normal(init.shape) * 0.01
    if perturb and name.endswith('gamma:0'):
        return init + 0.5 + rng.random(init.shape)
    if perturb and (name.endswith('beta:0') or name.endswith('bias:0')):
        return init + 0.2 * rng.standard_normal(init.shape)
    return init.copy()


def structured_images(batch, size, views, seed):
    """[batch, size, size, 3 * views] float64 in [0, 1]: images that differ from each other the way photographs do (a base colour per
    image and view plus a few low-frequency waves per channel and a little noise).  i.i.d. uniform noise makes every image statistically
    identical; an untrained deep network then maps the batch to one feature and every BatchNorm over the batch amplifies rounding
    noise -- useless for judging precision."""
    rng = np.random.default_rng([seed, 77])
    lin = np.linspace(0.0, 1.0, size)
    yy, xx = np.meshgrid(lin, lin, indexing='ij')
    C = 3 * views
    img = np.broadcast_to(0.2 + 0.6 * rng.random((batch, 1, 1, C)), (batch, size, size, C)).copy()
    for _ in range(4):
        fy, fx = 6.0 * rng.random((batch, 1, 1, C)), 6.0 * rng.random((batch, 1, 1, C))
        ph, amp = 2.0 * math.pi * rng.random((batch, 1, 1, C)), 0.25 * rng.random((batch, 1, 1, C))
        img += amp * np.sin(2.0 * math.pi * (fy * yy[None, :, :, None] + fx * xx[None, :, :, None]) + ph)
    img += 0.05 * rng.random((batch, size, size, C))
    return np.clip(img, 0.0, 1.0)


def one_hot_labels(batch, classes, seed):
    rng = np.random.default_rng([seed, 78])
    return np.eye(classes)[rng.integers(0, classes, batch)]


# the `*_img` cases: (tag -> configuration).  Batches large enough, and inputs image-like enough, that fp32 arithmetic itself stays an
# order of magnitude inside north_star's tolerances (torch-CPU fp32 vs float64 on these cases: embeddings 1e-6 ... 2e-6)
IMG_CASES = {
    'r18_img': dict(depth=18, size=32, batch=16, classes=10, perturb=True, seed=5),
    'r50_img': dict(depth=50, size=64, batch=16, classes=10, perturb=False, seed=6),
}
