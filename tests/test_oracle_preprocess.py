"""CPU: the pre-process restatement (oracle/preprocess.py) against analytic identities -- cv2 is absent,
so these pin the fixed-point warp's conventions (pixel centres, rounding, border) instead."""
import numpy as np

from oracle import preprocess as opre


def _img(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def test_identity_warp_returns_the_image():
    img = _img(40, 56)
    M = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(opre.warp_affine(img, M, (56, 40)), img)


def test_integer_shift_and_constant_border():
    img = _img(32, 48, 1)
    M = np.array([[1.0, 0, 5], [0, 1.0, -3]])          # dst(x, y) = src(x - 5, y + 3)
    out = opre.warp_affine(img, M, (48, 32))
    assert np.array_equal(out[:29, 5:], img[3:, :43])
    assert not out[:, :5].any() and not out[29:, :].any()


def test_half_pixel_shift_is_the_rounded_average():
    img = _img(16, 16, 2)
    M = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])          # src x = dst x - 0.5
    out = opre.warp_affine(img, M, (16, 16))
    want = (img[:, :-1].astype(np.int64) + img[:, 1:].astype(np.int64) + 1) >> 1     # (a + b + 1) >> 1 = (16384 a + 16384 b + 16384) >> 15
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))


def test_square_image_at_input_res_is_only_normalised():
    img = _img(64, 64, 3)
    inp, c, s = opre.get_input(img, res=64)
    want = ((img.astype(np.float32) / 255.) - opre.MEAN) / opre.STD
    assert np.array_equal(inp, want.transpose(2, 0, 1))
    assert c.tolist() == [32.0, 32.0] and s == 64.0


def test_landscape_image_is_letterboxed():
    img = _img(30, 60, 4) | 1                            # no zero pixels
    inp, c, s = opre.get_input(img, res=60)              # scale 1: 30 rows centred in 60
    raw = np.rint((inp.transpose(1, 2, 0) * opre.STD + opre.MEAN) * 255.).astype(np.int64)
    assert np.array_equal(raw[15:45], img.astype(np.int64))
    assert not raw[:15].any() and not raw[45:].any()


def test_general_affine_agrees_with_an_independent_float_bilinear_warp():
    # cv2 is absent; scipy's map_coordinates (float bilinear, constant border) is an INDEPENDENT implementation of the same
    # geometric map.  OpenCV's scheme quantises the sampling position to 1/32 px and the weights to 2^-15, so the two may
    # differ by a few intensity levels where the image is rough, never systematically: mean |diff| < 1 level, and the
    # pixel-centre / axis / rounding conventions (which a wrong restatement would shift by a whole pixel) agree.
    from scipy import ndimage
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(12, 16, 3)).astype(np.float64)
    img = np.clip(ndimage.zoom(base, (6, 6, 1), order=1), 0, 255).astype(np.uint8)           # smooth 72 x 96 image
    c = np.array([48.0, 36.0], np.float32)
    from oracle.post_process import get_affine_transform
    M = get_affine_transform(c, 130.0, 0, [64, 64])                                            # src -> dst, scale 64/130
    out = opre.warp_affine(img, M, (64, 64)).astype(np.float64)
    Mi = opre.invert_affine(M).reshape(2, 3)
    ys, xs = np.mgrid[0:64, 0:64].astype(np.float64)
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    ref = np.stack([ndimage.map_coordinates(img[..., ch].astype(np.float64), [sy, sx], order=1, mode="constant", cval=0.0)
                    for ch in range(3)], -1)
    inside = (sx >= 1) & (sx <= img.shape[1] - 2) & (sy >= 1) & (sy <= img.shape[0] - 2)
    d = np.abs(out - ref)[inside]
    assert d.mean() < 1.0 and d.max() <= 6.0, (d.mean(), d.max())
    # a one-pixel shift of the restatement's conventions would be far outside that
    shifted = np.abs(out[:, 1:] - ref[:, :-1])[inside[:, 1:]]
    assert shifted.mean() > 3 * d.mean()
