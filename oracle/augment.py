"""Oracle: the two-view augmentation pipeline (numpy float64).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/tf2/data_util.py:
  random_crop_with_resize / crop_and_resize / distorted_bounding_box_crop   :246-320, :362-377
  center_crop / _compute_crop_shape (eval path)                             :175-243
  tf.image.random_flip_left_right                                           :463
  random_color_jitter / color_jitter / color_jitter_rand / random_brightness / to_grayscale   :34-173, :380-389
  preprocess_for_train / preprocess_for_eval                                :443-499
and the two-views-per-image concat of /root/reference/tf2/data.py:52-62.

The pixel arithmetic of several steps lives in TensorFlow kernels that are NOT under /root/reference (third-party
dependency, unpinned `tensorflow` in tf2/requirements.txt); they are restated here from TensorFlow's published
sources / API documentation and flagged "TF semantics" -- not verifiable in this image (PARITY UNPINNED):
  * tf.image.sample_distorted_bounding_box  (core/kernels/image/sample_distorted_bounding_box_op.cc, GenerateRandomCrop)
  * tf.image.resize(..., BICUBIC) of TF2 = ResizeBicubic with half_pixel_centers=True: Keys cubic, A = -0.5,
    weights from a 1024-entry table, taps that fall outside the image dropped and the rest renormalised
    (core/kernels/image/resize_bicubic_op.cc)
  * tf.image.adjust_contrast ((x - mean_HW) * f + mean_HW per channel), adjust_saturation / adjust_hue
    (RGB -> HSV, scale + clip S / shift H mod 1, HSV -> RGB), rgb_to_grayscale (weights 0.2989, 0.5870, 0.1140)
  * tf.image.convert_image_dtype(uint8 -> float32) = x * (1 / 255)
Random DRAWS are distributional (any generator); the arithmetic given the draws is what the HIP kernels are
checked against (tests/gpu_checks.py::check_augment).
"""
import numpy as np

K_TABLE = 1024  # resize_bicubic_op.cc: kTableSize = 1 << 10


# ------------------------------------------------------------------ crop sampling (TF semantics)
def _lrint(x):
    return int(np.rint(x))            # round half to even, like lrintf under the default rounding mode


def generate_random_crop(rng, width, height, min_rel_area, max_rel_area, aspect_ratio):
    """GenerateRandomCrop of sample_distorted_bounding_box_op.cc.  Returns (x, y, w, h) or None."""
    if max_rel_area <= 0.0 or aspect_ratio <= 0.0 or width <= 0 or height <= 0 or min_rel_area > max_rel_area:
        return None
    f32 = np.float32
    min_area = f32(min_rel_area) * f32(width) * f32(height)
    max_area = f32(max_rel_area) * f32(width) * f32(height)
    h = _lrint(np.sqrt(f32(min_area / f32(aspect_ratio))))
    max_h = _lrint(np.sqrt(f32(max_area / f32(aspect_ratio))))
    if _lrint(max_h * aspect_ratio) > width:
        eps = 0.0000001
        max_h = int((width + 0.5 - eps) / aspect_ratio)
        if _lrint(max_h * aspect_ratio) > width:
            max_h -= 1
    max_h = min(max_h, height)
    h = min(h, max_h)
    if h < max_h:
        h += int(rng.integers(0, max_h - h + 1))       # random->Uniform(n): integer in [0, n)
    w = _lrint(h * aspect_ratio)
    area = float(w * h)
    if area < min_area:
        h += 1
        w = _lrint(h * aspect_ratio)
        area = float(w * h)
    if area > max_area:
        h -= 1
        w = _lrint(h * aspect_ratio)
        area = float(w * h)
    if area < min_area or area > max_area or w > width or h > height or w <= 0 or h <= 0:
        return None
    y = int(rng.integers(0, height - h)) if h < height else 0
    x = int(rng.integers(0, width - w)) if w < width else 0
    return x, y, w, h


def sample_distorted_bounding_box(rng, height, width, min_object_covered=0.1, aspect_ratio_range=(0.75, 1.33),
                                  area_range=(0.05, 1.0), max_attempts=100):
    """tf.image.sample_distorted_bounding_box with the single whole-image box that crop_and_resize passes
    (tf2/data_util.py:310-320).  The sampled crop must cover >= min_object_covered of that box, i.e. of the image.
    Returns (offset_y, offset_x, target_height, target_width); the whole image after max_attempts failures."""
    for _ in range(max_attempts):
        ar = float(np.float32(rng.random()) * np.float32(aspect_ratio_range[1] - aspect_ratio_range[0]) + np.float32(aspect_ratio_range[0]))
        c = generate_random_crop(rng, width, height, area_range[0], area_range[1], ar)
        if c is None:
            continue
        x, y, w, h = c
        if (w * h) / float(width * height) >= min_object_covered:
            return y, x, h, w
    return 0, 0, height, width


def compute_crop_shape(image_height, image_width, aspect_ratio, crop_proportion):
    """tf2/data_util.py:175-213 (tf.math.rint = round half to even)."""
    iw, ih = np.float32(image_width), np.float32(image_height)
    if aspect_ratio > iw / ih:
        ch = _lrint(np.float32(crop_proportion / aspect_ratio) * iw)
        cw = _lrint(np.float32(crop_proportion) * iw)
    else:
        ch = _lrint(np.float32(crop_proportion) * ih)
        cw = _lrint(np.float32(crop_proportion * aspect_ratio) * ih)
    return ch, cw


def center_crop_box(image_height, image_width, height, width, crop_proportion):
    """tf2/data_util.py:216-243: (offset_y, offset_x, crop_height, crop_width)."""
    ch, cw = compute_crop_shape(image_height, image_width, width / height, crop_proportion)
    return ((image_height - ch) + 1) // 2, ((image_width - cw) + 1) // 2, ch, cw


# ------------------------------------------------------------------ bicubic resize (TF semantics)
def _cubic_tables(a=-0.5):
    x = np.arange(K_TABLE + 1, dtype=np.float64) / K_TABLE
    t0 = ((a + 2) * x - (a + 3)) * x * x + 1            # |x| <= 1
    x1 = x + 1.0
    t1 = ((a * x1 - 5 * a) * x1 + 8 * a) * x1 - 4 * a   # 1 < |x| < 2
    # the kernel keeps its table in float32
    return t0.astype(np.float32).astype(np.float64), t1.astype(np.float32).astype(np.float64)


_T0, _T1 = _cubic_tables()


def bicubic_taps(out_size, in_size):
    """Per output coordinate: 4 source indices and 4 weights (HalfPixelScaler + GetWeightsAndIndices<.., use_keys_cubic>)."""
    scale = np.float32(in_size) / np.float32(out_size)
    idx = np.zeros((out_size, 4), dtype=np.int64)
    wts = np.zeros((out_size, 4), dtype=np.float64)
    for o in range(out_size):
        in_loc_f = np.float32((np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5))
        in_loc = int(np.floor(in_loc_f))
        delta = np.float32(in_loc_f - np.float32(in_loc))
        off = _lrint(delta * np.float32(K_TABLE))
        w = [_T1[off], _T0[off], _T0[K_TABLE - off], _T1[K_TABLE - off]]
        ii = []
        for j in range(4):
            want = in_loc - 1 + j
            got = min(max(want, 0), in_size - 1)
            if got != want:
                w[j] = 0.0                                # a tap outside the image contributes nothing ...
            ii.append(got)
        s = float(np.float32(w[0]) + np.float32(w[1]) + np.float32(w[2]) + np.float32(w[3]))
        if abs(s) >= 1000.0 * np.finfo(np.float32).tiny:  # ... and the remaining weights are renormalised
            w = [wj / s for wj in w]
        idx[o], wts[o] = ii, w
    return idx, wts


def resize_bicubic(img, out_h, out_w):
    """img [h, w, c] float -> [out_h, out_w, c]: separable, rows then columns (the kernel's order of accumulation differs
    only in rounding)."""
    img = np.asarray(img, dtype=np.float64)
    iy, wy = bicubic_taps(out_h, img.shape[0])
    ix, wx = bicubic_taps(out_w, img.shape[1])
    rows = (img[iy] * wy[:, :, None, None]).sum(1)                    # [out_h, w, c]
    return (rows[:, ix] * wx[None, :, :, None]).sum(2)                # [out_h, out_w, c]


# ------------------------------------------------------------------ colour ops (TF semantics)
def rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = np.maximum(r, np.maximum(g, b))
    rng_ = v - np.minimum(r, np.minimum(g, b))
    s = np.where(v > 0, rng_ / np.where(v > 0, v, 1.0), 0.0)
    norm = 1.0 / (6.0 * np.where(rng_ > 0, rng_, 1.0))
    h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
    h = np.where(rng_ <= 0, 0.0, h)
    h = np.where(h < 0, h + 1.0, h)
    return np.stack([h, s, v], -1)


def hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    c = s * v
    m = v - c
    dh = h * 6.0
    cat = np.floor(dh).astype(np.int64)
    fm = np.mod(dh, 2.0)
    x = c * (1.0 - np.abs(fm - 1.0))
    z = np.zeros_like(c)
    rr = np.choose(np.clip(cat, 0, 5), [c, x, z, z, x, c])
    gg = np.choose(np.clip(cat, 0, 5), [x, c, c, x, z, z])
    bb = np.choose(np.clip(cat, 0, 5), [z, z, x, c, c, x])
    return np.stack([rr + m, gg + m, bb + m], -1)


def adjust_brightness_mul(img, factor):          # random_brightness impl='simclrv2', tf2/data_util.py:34-45
    return img * factor


def adjust_contrast(img, factor):                # tf.image.adjust_contrast
    mean = img.mean((0, 1), keepdims=True)
    return (img - mean) * factor + mean


def adjust_saturation(img, factor):              # tf.image.adjust_saturation
    hsv = rgb_to_hsv(img)
    hsv[..., 1] = np.clip(hsv[..., 1] * factor, 0.0, 1.0)
    return hsv_to_rgb(hsv)


def adjust_hue(img, delta):                      # tf.image.adjust_hue
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = np.mod(hsv[..., 0] + delta + 1.0, 1.0)
    return hsv_to_rgb(hsv)


def to_grayscale(img):                           # tf2/data_util.py:48-52
    g = img[..., 0] * 0.2989 + img[..., 1] * 0.5870 + img[..., 2] * 0.1140
    return np.repeat(g[..., None], 3, -1)


def color_jitter_given(img, perm, brightness_f, contrast_f, saturation_f, hue_delta):
    """color_jitter_rand (tf2/data_util.py:122-173) with the draws given: op i of `perm` then clip to [0,1], four times."""
    for i in perm:
        if i == 0:
            img = adjust_brightness_mul(img, brightness_f)
        elif i == 1:
            img = adjust_contrast(img, contrast_f)
        elif i == 2:
            img = adjust_saturation(img, saturation_f)
        else:
            img = adjust_hue(img, hue_delta)
        img = np.clip(img, 0.0, 1.0)
    return img


# ------------------------------------------------------------------ parameter draws + the pipeline
PARAM_FIELDS = ('crop_y', 'crop_x', 'crop_h', 'crop_w', 'flip', 'jitter_on', 'perm0', 'perm1', 'perm2', 'perm3',
                'brightness', 'contrast', 'saturation', 'hue', 'gray_on', 'pad')


def draw_train_params(rng, src_h, src_w, height, width, color_jitter_strength=1.0, crop=True, flip=True):
    """One view's random draws, in the order preprocess_for_train consumes them (tf2/data_util.py:443-475).
    Returns a float64 vector laid out as PARAM_FIELDS."""
    p = np.zeros(len(PARAM_FIELDS))
    if crop:                                                                       # :362-377, :298-320
        ar = width / height
        y, x, h, w = sample_distorted_bounding_box(rng, src_h, src_w, 0.1, (3. / 4 * ar, 4. / 3. * ar), (0.08, 1.0), 100)
    else:
        y, x, h, w = 0, 0, src_h, src_w
    p[0:4] = (y, x, h, w)
    p[4] = float(flip and rng.random() < 0.5)                                      # :463
    s = color_jitter_strength
    if s > 0:                                                                      # :380-389
        p[5] = float(rng.random() < 0.8)
        p[6:10] = rng.permutation(4)                                               # :168
        b, c, sa, hu = 0.8 * s, 0.8 * s, 0.8 * s, 0.2 * s                          # :70-73
        p[10] = rng.uniform(max(1.0 - b, 0.0), 1.0 + b)                            # :37-39
        p[11] = rng.uniform(1 - c, 1 + c)
        p[12] = rng.uniform(1 - sa, 1 + sa)
        p[13] = rng.uniform(-hu, hu)
        p[14] = float(rng.random() < 0.2)
    return p


def apply_train_params(image, p, height, width):
    """preprocess_for_train given the draws.  image: [h, w, 3] uint8 or float in [0,1]."""
    img = np.asarray(image)
    img = img.astype(np.float64) * (1.0 / 255.0) if img.dtype == np.uint8 else img.astype(np.float64)
    y, x, h, w = (int(v) for v in p[0:4])
    if (h, w) != img.shape[:2] or True:
        img = resize_bicubic(img[y:y + h, x:x + w], height, width)               # crop_to_bounding_box + resize BICUBIC
    if p[4] > 0:
        img = img[:, ::-1]
    if p[5] > 0:
        img = color_jitter_given(img, [int(v) for v in p[6:10]], p[10], p[11], p[12], p[13])
    if p[14] > 0:
        img = to_grayscale(img)
    return np.clip(img, 0.0, 1.0)                                                  # :473-474


def two_view_batch(images, params, height, width):
    """tf2/data.py:52-62: two transformations of every image, concatenated on the channel axis -> [b, H, W, 6].
    images: list of [h_i, w_i, 3]; params: [b, 2, len(PARAM_FIELDS)]."""
    out = np.zeros((len(images), height, width, 6))
    for i, im in enumerate(images):
        for v in range(2):
            out[i, :, :, 3 * v:3 * v + 3] = apply_train_params(im, params[i, v], height, width)
    return out


def preprocess_for_eval(image, height, width, crop=True, crop_proportion=0.875):
    """tf2/data_util.py:478-499 (CROP_PROPORTION = 0.875, :27)."""
    img = np.asarray(image)
    img = img.astype(np.float64) * (1.0 / 255.0) if img.dtype == np.uint8 else img.astype(np.float64)
    if crop:
        y, x, h, w = center_crop_box(img.shape[0], img.shape[1], height, width, crop_proportion)
        img = resize_bicubic(img[y:y + h, x:x + w], height, width)
    return np.clip(img, 0.0, 1.0)
