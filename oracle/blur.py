"""Oracle: on-device Gaussian blur augmentation (numpy float64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED TO THE REFERENCE'S SOURCE (tests/golden/reference_pin.npz: tf2/*.py executed on oracle/tfshim.py); TensorFlow's own kernels unpinned.
Restates /root/reference/tf2/data_util.py:323-361 (gaussian_blur) and :413-440 (batch_random_blur)
with the random draws (sigma per view, selector per image) passed in explicitly.
"""
import numpy as np


def gaussian_blur(image, kernel_size, sigma):
    """image [b,H,W,C]; separable depthwise conv, padding='SAME' (zeros), horizontal then vertical."""
    radius = int(kernel_size / 2)                      # :338
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    f = np.exp(-x ** 2 / (2.0 * float(sigma) ** 2))    # :341-342
    f = f / f.sum()                                    # :343
    b, H, W, C = image.shape
    pad = np.zeros((b, H, W + 2 * radius, C)); pad[:, :, radius:radius + W] = image
    hb = sum(f[t] * pad[:, :, t:t + W] for t in range(2 * radius + 1))        # blur_h :355-356
    pad = np.zeros((b, H + 2 * radius, W, C)); pad[:, radius:radius + H] = hb
    return sum(f[t] * pad[:, t:t + H] for t in range(2 * radius + 1))         # blur_v :357-358


def batch_random_blur(images_list, height, sigmas, selectors):
    """:431-438 with explicit draws: images_new*selector + images*(1-selector), clipped to [0,1]."""
    out = []
    for images, sigma, sel in zip(images_list, sigmas, selectors):
        images = np.asarray(images, dtype=np.float64)
        new = gaussian_blur(images, height // 10, sigma)                      # random_blur :406-407
        s = np.asarray(sel, dtype=np.float64).reshape(-1, 1, 1, 1)
        out.append(np.clip(new * s + images * (1 - s), 0.0, 1.0))
    return out
