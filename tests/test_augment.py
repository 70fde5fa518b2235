"""CPU tests of the augmentation oracle (oracle/augment.py) and of the host-side parameter draws
(simclr_amd/data_util.py): closed-form properties of the restated TensorFlow ops and distributional agreement of the
vectorised crop sampler with the scalar restatement of tf.image.sample_distorted_bounding_box."""
import numpy as np

from oracle import augment as oa
from simclr_amd import data_util as du


def test_bicubic_identity_constant_and_linear_reproduction():
    rng = np.random.default_rng(0)
    img = rng.random((9, 7, 3))
    assert np.abs(oa.resize_bicubic(img, 9, 7) - img).max() < 1e-12          # same size: delta = 0 -> weights (0,1,0,0)
    const = np.full((5, 6, 3), 0.37)
    assert np.abs(oa.resize_bicubic(const, 11, 13) - 0.37).max() < 1e-6       # weights renormalised to sum 1
    # Keys cubic convolution reproduces linear ramps away from the borders (up to the 1/1024 weight table)
    ramp = np.tile(np.arange(32, dtype=np.float64)[None, :, None], (4, 1, 3)) / 32
    out = oa.resize_bicubic(ramp, 4, 64)
    xs = ((np.arange(64) + 0.5) * 0.5 - 0.5) / 32
    assert np.abs(out[0, 8:-8, 0] - xs[8:-8]).max() < 2e-3


def test_bicubic_taps_half_pixel_centres():
    idx, w = oa.bicubic_taps(4, 8)          # scale 2: out 0 samples in_loc_f = 0.5 -> taps (-1,0,1,2) -> (0,0,1,2), first dropped
    assert idx[0].tolist() == [0, 0, 1, 2] and w[0][0] == 0.0 and abs(w[0].sum() - 1) < 1e-6
    assert idx[1].tolist() == [1, 2, 3, 4] and abs(w[1][1] - w[1][2]) < 1e-6   # delta = 0.5: symmetric


def test_colour_ops_identities_and_known_values():
    rng = np.random.default_rng(1)
    img = rng.random((6, 5, 3))
    assert np.abs(oa.hsv_to_rgb(oa.rgb_to_hsv(img)) - img).max() < 1e-12
    assert np.abs(oa.adjust_contrast(img, 1.0) - img).max() < 1e-12
    assert np.abs(oa.adjust_saturation(img, 1.0) - img).max() < 1e-12
    assert np.abs(oa.adjust_hue(img, 0.0) - img).max() < 1e-12
    assert np.abs(oa.adjust_hue(oa.adjust_hue(img, 0.3), -0.3) - img).max() < 1e-12
    # pure red: h=0, s=1, v=1; hue +1/3 -> pure green; saturation 0 -> gray of value v
    red = np.array([[[1.0, 0.0, 0.0]]])
    assert np.abs(oa.adjust_hue(red, 1.0 / 3) - [[[0.0, 1.0, 0.0]]]).max() < 1e-12
    assert np.abs(oa.adjust_saturation(np.array([[[0.8, 0.2, 0.4]]]), 0.0) - 0.8).max() < 1e-12
    assert np.abs(oa.to_grayscale(red) - 0.2989).max() < 1e-12
    c = oa.adjust_contrast(np.array([[[0.0, 0.5, 1.0]], [[1.0, 0.5, 0.0]]]), 2.0)   # per-channel mean 0.5
    assert np.allclose(c[:, 0, 0], [-0.5, 1.5]) and np.allclose(c[:, 0, 1], [0.5, 0.5])


def test_center_crop_box_and_eval_preprocess():
    assert oa.center_crop_box(256, 256, 224, 224, 0.875) == (16, 16, 224, 224)
    assert oa.center_crop_box(300, 500, 224, 224, 0.875) == (19, 119, 262, 262)    # image wider than the target
    img = np.random.default_rng(2).integers(0, 256, (64, 80, 3), dtype=np.uint8)
    out = oa.preprocess_for_eval(img, 32, 32)
    assert out.shape == (32, 32, 3) and out.min() >= 0 and out.max() <= 1
    b = du.center_crop_boxes([256, 300], [256, 500], 224, 224)
    assert b.tolist() == [[16, 16, 224, 224], [19, 119, 262, 262]]


def test_crop_sampler_constraints_and_agreement_with_scalar_restatement():
    rng = np.random.default_rng(3)
    H, W = 300, 400
    box = du.sample_crop_boxes(rng, np.full(6000, H), np.full(6000, W), 224, 224)
    y, x, h, w = box.T
    assert (y >= 0).all() and (x >= 0).all() and (y + h <= H).all() and (x + w <= W).all()
    area = h * w / (H * W)
    assert area.min() >= 0.1 - 1e-9 and area.max() <= 1.0          # min_object_covered 0.1 beats area_range's 0.08
    ar = w / h
    assert ar.min() > 0.74 and ar.max() < 1.35
    rng2 = np.random.default_rng(4)
    ref = np.array([oa.sample_distorted_bounding_box(rng2, H, W, 0.1, (0.75, 4. / 3), (0.08, 1.0)) for _ in range(6000)])
    a2 = ref[:, 2] * ref[:, 3] / (H * W)
    for q in (0.1, 0.25, 0.5, 0.75, 0.9):
        assert abs(np.quantile(area, q) - np.quantile(a2, q)) < 0.03
    assert abs(np.mean(y / np.maximum(H - h, 1)) - 0.5) < 0.03     # offsets uniform over the admissible range


def test_train_param_draw_probabilities_and_ranges():
    p = du.draw_train_params(4000, 200, 260, 64, 64, 1.0, rng=np.random.default_rng(5)).reshape(-1, 16)
    assert abs(p[:, 4].mean() - 0.5) < 0.03 and abs(p[:, 5].mean() - 0.8) < 0.03 and abs(p[:, 14].mean() - 0.2) < 0.03
    assert (np.sort(p[:, 6:10], 1) == np.arange(4)).all()
    assert p[:, 10].min() >= 0.2 - 1e-6 and p[:, 10].max() <= 1.8 + 1e-6       # brightness factor, strength 1
    assert np.abs(p[:, 13]).max() <= 0.2 + 1e-6
    first = np.bincount(p[:, 6].astype(int), minlength=4) / len(p)
    assert np.abs(first - 0.25).max() < 0.03                                     # uniform random order
    p0 = du.draw_train_params(8, 50, 60, 32, 32, 0.0, crop=False, flip=False, rng=np.random.default_rng(6)).reshape(-1, 16)
    assert (p0[:, 0:4] == [0, 0, 50, 60]).all() and not p0[:, 4:].any()


def test_pipeline_given_draws_is_the_composition():
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    p = np.zeros(16)
    p[0:4] = (4, 6, 30, 36); p[4] = 1; p[5] = 1; p[6:10] = (2, 0, 3, 1); p[10:14] = (1.3, 0.7, 1.4, -0.1); p[14] = 1
    out = oa.apply_train_params(img, p, 16, 16)
    x = oa.resize_bicubic(img[4:34, 6:42].astype(np.float64) / 255.0, 16, 16)[:, ::-1]
    x = np.clip(oa.adjust_saturation(x, 1.4), 0, 1)
    x = np.clip(x * 1.3, 0, 1)
    x = np.clip(oa.adjust_hue(x, -0.1), 0, 1)
    x = np.clip(oa.adjust_contrast(x, 0.7), 0, 1)
    assert np.abs(out - np.clip(oa.to_grayscale(x), 0, 1)).max() < 1e-12
