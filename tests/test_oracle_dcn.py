"""CPU: pin the DCNv2 restatements (oracle/dcn.py torch, oracle/dcn_ref.c plain C) with the
reference's own known-answer test (DCNv2/test.py:32-67: zero offset + mask 0.5 + identity
kernel => input == 2*output) and derived identities."""
import numpy as np
import torch
import torch.nn.functional as F

import h3d_amd  # noqa: F401
from h3d_amd import synth
from oracle import dcn as odcn


def _rand(key, shape, lo=-1.0, hi=1.0, seed=0):
    return torch.from_numpy(synth.uniform(key, shape, lo, hi, seed))


def _both(x, w, b, off, m, *a):
    y_t = odcn.dcn_v2_forward(x, w, b, off, m, *a)
    y_c = torch.from_numpy(odcn.dcn_v2_forward_c(x.numpy(), w.numpy(), b.numpy(), off.numpy(),
                                                 m.numpy(), *a))
    np.testing.assert_allclose(y_t.numpy(), y_c.numpy(), rtol=2e-5, atol=2e-5)
    return y_t, y_c


def test_zero_offset_identity_reference_known_answer():
    # DCNv2/test.py:32-67, shapes N,C,H,W = 2,2,4,4
    N, C, H, W = 2, 2, 4, 4
    x = _rand("x", (N, C, H, W))
    w = torch.zeros(C, C, 3, 3)
    for p in range(C):
        w[p, p, 1, 1] = 1.0
    b = torch.zeros(C)
    off = torch.zeros(N, 18, H, W)
    m = torch.full((N, 9, H, W), 0.5)
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        assert (x - 2 * y).abs().max().item() < 1e-10


def test_zero_offset_unit_mask_is_conv2d():
    x = _rand("x", (2, 6, 9, 7))
    w = _rand("w", (5, 6, 3, 3), -0.3, 0.3)
    b = _rand("b", (5,))
    off = torch.zeros(2, 18, 9, 7)
    m = torch.ones(2, 9, 9, 7)
    ref = F.conv2d(x, w, b, 1, 1)
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_fresh_dcn_module_is_half_conv():
    # init_offset zeroes conv_offset_mask (dcn_v2.py:114-116): offsets 0, mask sigmoid(0)=0.5
    x = _rand("x", (1, 4, 8, 8))
    w = _rand("w", (3, 4, 3, 3), -0.3, 0.3)
    b = _rand("b", (3,))
    y = odcn.dcn_module_forward(x, w, b, torch.zeros(27, 4, 3, 3), torch.zeros(27))
    np.testing.assert_allclose(y.numpy(), (0.5 * F.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1)).numpy(),
                               rtol=1e-5, atol=1e-5)


def test_integer_offsets_are_a_shifted_conv():
    x = _rand("x", (1, 3, 10, 10))
    w = _rand("w", (2, 3, 3, 3), -0.3, 0.3)
    b = torch.zeros(2)
    off = torch.zeros(1, 18, 10, 10)
    off[:, 0::2] = 1.0      # every tap one row down
    off[:, 1::2] = -2.0     # and two columns left
    m = torch.ones(1, 9, 10, 10)
    xp = F.pad(x, (3, 3, 3, 3))
    shifted = xp[:, :, 3 + 1:3 + 1 + 10, 3 - 2:3 - 2 + 10]
    # shifted conv with zero padding of the ORIGINAL image: build via unfold on padded input
    ref = F.conv2d(F.pad(x, (4, 4, 4, 4)), w, b)[:, :, 1 + 3:1 + 3 + 10, 3 - 2:3 - 2 + 10]
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert shifted.shape == x.shape


def test_far_offsets_give_bias_only():
    x = _rand("x", (1, 2, 5, 5))
    w = _rand("w", (3, 2, 3, 3))
    b = _rand("b", (3,))
    off = torch.full((1, 18, 5, 5), 100.0)
    m = torch.ones(1, 9, 5, 5)
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        np.testing.assert_allclose(y.numpy(), b.view(1, 3, 1, 1).expand(1, 3, 5, 5).numpy(), atol=1e-7)


def test_boundary_gate_and_zero_corners():
    # a sample at h_im = -0.5 is inside the (> -1) gate; its low corner row is outside -> 0
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0
    b = torch.zeros(1)
    off = torch.zeros(1, 18, 4, 4)
    off[0, 2 * 4, 0, :] = -0.5            # centre tap, row 0: h_im = -0.5
    m = torch.ones(1, 9, 4, 4)
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        np.testing.assert_allclose(y[0, 0, 0].numpy(), np.full(4, 0.5, np.float32), atol=1e-7)
        np.testing.assert_allclose(y[0, 0, 1:].numpy(), np.ones((3, 4), np.float32), atol=1e-7)
    off[0, 2 * 4, 0, :] = -1.0            # exactly -1: gate closed (h_im > -1 is false)
    for y in _both(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1):
        np.testing.assert_allclose(y[0, 0, 0].numpy(), np.zeros(4, np.float32), atol=1e-7)


def test_random_offsets_stride_dilation_groups_c_vs_torch():
    B, C, H, W, Co, dg = 2, 8, 11, 9, 6, 2
    for (s, p, d) in [(1, 1, 1), (2, 1, 1), (1, 2, 2)]:
        Ho = (H + 2 * p - (d * 2 + 1)) // s + 1
        Wo = (W + 2 * p - (d * 2 + 1)) // s + 1
        x = _rand("x", (B, C, H, W))
        w = _rand("w", (Co, C, 3, 3), -0.3, 0.3)
        b = _rand("b", (Co,))
        off = _rand("off", (B, 18 * dg, Ho, Wo), -3.0, 3.0)
        m = _rand("m", (B, 9 * dg, Ho, Wo), 0.0, 1.0)
        y_t, y_c = _both(x, w, b, off, m, 3, 3, s, s, p, p, d, d, dg)
        assert y_t.shape == (B, Co, Ho, Wo)
        y64 = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, s, s, p, p, d, d, dg, acc_dtype=torch.float64)
        np.testing.assert_allclose(y64.numpy(), y_c.numpy(), rtol=2e-6, atol=2e-6)


def test_shape_smoke_dg2():
    # DCNv2/test.py:169-180 example shape class (smaller spatial size to stay fast)
    x = _rand("x", (2, 64, 16, 16))
    w = _rand("w", (64, 64, 3, 3), -0.05, 0.05)
    y = odcn.dcn_v2_forward(x, w, torch.zeros(64), torch.zeros(2, 36, 16, 16),
                            torch.ones(2, 18, 16, 16), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    assert y.shape == (2, 64, 16, 16)


def test_sampling_rule_agrees_with_scipy_map_coordinates():
    """The reference's CUDA DCNv2 cannot be built here, and it has no CPU path: besides the reference's own known answer and the
    derived identities above, the restatement's whole sampling rule -- bilinear weights from floor(), corners outside the
    image contribute zero, samples at h_im <= -1 or >= H vanish (dcn_v2_im2col_cuda.cu:25-54, 163-185) -- is cross-checked
    against an INDEPENDENT implementation: scipy.ndimage.map_coordinates(order=1, mode='grid-constant', cval=0) is exactly
    "bilinear interpolation of the zero-extended image" (scipy's plain 'constant' does not blend across the edge).  Offsets are large enough to leave the image on every side."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    B, C, H, W, Co = 2, 3, 9, 11, 4
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((Co, C, 3, 3)).astype(np.float32) * 0.3
    b = rng.standard_normal(Co).astype(np.float32)
    off = rng.uniform(-4.0, 4.0, (B, 18, H, W)).astype(np.float32)
    m = rng.uniform(0.0, 1.0, (B, 9, H, W)).astype(np.float32)
    got = odcn.dcn_v2_forward(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), torch.from_numpy(off),
                              torch.from_numpy(m), 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64).numpy()
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    want = np.zeros((B, Co, H, W))
    for bi in range(B):
        for t in range(9):
            i, j = divmod(t, 3)
            py = ys - 1 + i + off[bi, 2 * t].astype(np.float64)          # channel 2t = dh, 2t+1 = dw (im2col.cu:170-171)
            px = xs - 1 + j + off[bi, 2 * t + 1].astype(np.float64)
            for c in range(C):
                s = ndimage.map_coordinates(x[bi, c].astype(np.float64), [py, px], order=1, mode="grid-constant", cval=0.0)
                want[bi] += w[:, c, i, j][:, None, None] * (s * m[bi, t])[None]
        want[bi] += b[:, None, None]
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_f16_blend_mode_models_four_fp16_roundings():
    """oracle/dcn.py blend='f16' (round 5: the 2-byte GPU plans' blend, csrc/dcn_traits.h SE<bf16_t>::blend): with weights and samples
    that are exactly representable and whose products need no rounding it equals the fp32 blend bit for bit; in general it differs
    from it by at most four fp16 roundings of the running sum (2^-11 relative each)."""
    torch.manual_seed(0)
    B, C, H, W, Co = 1, 4, 7, 9, 3
    x = torch.randint(-8, 9, (B, C, H, W)).float()                      # small integers: exact in fp16, products exact
    w = torch.randint(-2, 3, (Co, C, 3, 3)).float()
    b = torch.zeros(Co)
    off = torch.zeros(B, 18, H, W)
    off[:, 0::2] = 0.5                                                  # half-pixel shifts: bilinear weights 0.25 / 0.5, exact
    off[:, 1::2] = -0.5
    m = torch.full((B, 9, H, W), 0.5)
    a = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
    f = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64, blend="f16")
    assert torch.equal(a, f)
    x = torch.randn(B, C, H, W).half().float()
    off = torch.randn(B, 18, H, W) * 1.5
    m = torch.rand(B, 9, H, W)
    w = torch.randn(Co, C, 3, 3).half().float() * 0.2
    a = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
    f = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64, blend="f16")
    scale = float(a.abs().max())
    err = float((a - f).abs().max())
    assert 0 < err < 6 * 2.0 ** -11 * scale, (err, scale)
