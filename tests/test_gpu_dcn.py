"""GPU parity of the deformable-conv kernels against the oracle (oracle/dcn.py, pinned in
tests/test_oracle_dcn.py): (1) the operator boundary `dcn_v2_forward` (NCHW fp32, general
parameters), incl. the reference's own known-answer test DCNv2/test.py:32-67; (2) the network
DCN op (NHWC, fused offset/mask layout) in f32 and bf16."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import DEV, TD, bf16_round, from_nhwc, mk, nhwc, pack_conv, rnd, run
from h3d_amd import _lib, dcn_v2
from oracle import dcn as odcn

pytestmark = pytest.mark.gpu


def _fwd(x, w, b, off, m, *a):
    with torch.no_grad():
        return dcn_v2.dcn_v2_forward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), m.to(DEV), *a).cpu()


def test_zero_offset_identity_reference_known_answer():
    N, C, H, W = 2, 2, 4, 4
    x = rnd("x", (N, C, H, W))
    w = torch.zeros(C, C, 3, 3)
    for p in range(C):
        w[p, p, 1, 1] = 1.0
    y = _fwd(x, w, torch.zeros(C), torch.zeros(N, 18, H, W), torch.full((N, 9, H, W), 0.5), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert (x - 2 * y).abs().max().item() < 1e-10


def test_shape_smoke_dg2_reference_example():
    # DCNv2/test.py:169-180: DCN(64, 64, 3x3, dg=2) on [2,64,128,128]
    x = rnd("x", (2, 64, 128, 128))
    w = rnd("w", (64, 64, 3, 3), -0.05, 0.05)
    y = _fwd(x, w, torch.zeros(64), torch.zeros(2, 36, 128, 128), torch.ones(2, 18, 128, 128), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    assert y.shape == (2, 64, 128, 128)
    ref = F.conv2d(x, w, None, 1, 1)
    assert float((y - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("cfg", [(1, 1, 1, 1), (2, 1, 1, 1), (1, 2, 2, 2), (1, 1, 1, 4)])
def test_operator_random_vs_oracle(cfg):
    s, p, d, dg = cfg
    B, C, H, W, Co = 2, 8, 13, 11, 7
    Ho = (H + 2 * p - (d * 2 + 1)) // s + 1
    Wo = (W + 2 * p - (d * 2 + 1)) // s + 1
    x = rnd("x", (B, C, H, W))
    w = rnd("w", (Co, C, 3, 3), -0.3, 0.3)
    b = rnd("b", (Co,))
    off = rnd("off", (B, 18 * dg, Ho, Wo), -3.0, 3.0)
    m = rnd("m", (B, 9 * dg, Ho, Wo), 0.0, 1.0)
    y = _fwd(x, w, b, off, m, 3, 3, s, s, p, p, d, d, dg)
    ref = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, s, s, p, p, d, d, dg, acc_dtype=torch.float64)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 64, 64, 24, 40), (1, 128, 64, 20, 36), (1, 256, 128, 16, 16), (1, 48, 7, 13, 11), (1, 64, 200, 9, 9)])
@pytest.mark.parametrize("oscale", [1.0, 6.0])
def test_operator_fast_path_matches_oracle_and_general_kernel(shape, oscale):
    # the model's configuration (model.py:355: 3x3 s1 p1 d1 dg1, C % 16 == 0) goes through h3d_dcn_v2_forward_ws:
    # NCHW operands re-laid in a workspace, LDS-apron gather + fp32 MFMA (csrc/dcn2.hip); compared with the oracle
    # (fp64 accumulation) and with the general kernel behind h3d_dcn_v2_forward on the same operands
    import ctypes
    B, C, Co, H, W = shape
    x = rnd("x", (B, C, H, W))
    w = rnd("w", (Co, C, 3, 3)) * (1.5 / np.sqrt(C * 9))
    b = rnd("b", (Co,))
    off = rnd("off", (B, 18, H, W), -oscale, oscale)
    off[:, :, :2, :] *= 4.0                                  # some samples far outside the apron / the image
    m = rnd("m", (B, 9, H, W), 0.0, 1.0)
    y = _fwd(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    ref = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    xd, wd, bd, od, md = [t.contiguous().to(DEV) for t in (x, w, b, off, m)]
    gen = torch.empty(B, Co, H, W, device=DEV)
    L = _lib.lib()
    _lib.check(L.h3d_dcn_v2_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(od), _lib.ptr(md), _lib.ptr(gen), B, C, H, W,
                                    Co, 3, 3, 1, 1, 1, 1, 1, 1, 1, _lib.stream_ptr()), "general")
    np.testing.assert_allclose(y.numpy(), gen.cpu().numpy(), rtol=1e-5, atol=2e-5)
    # a short workspace falls back to the general kernel instead of overrunning it
    out2 = torch.empty_like(gen)
    ws = torch.empty(64, dtype=torch.uint8, device=DEV)
    _lib.check(L.h3d_dcn_v2_forward_ws(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(od), _lib.ptr(md), _lib.ptr(out2), B, C, H,
                                       W, Co, 3, 3, 1, 1, 1, 1, 1, 1, 1, _lib.ptr(ws), 64, _lib.stream_ptr()), "short ws")
    torch.cuda.synchronize()
    assert torch.equal(out2, gen)


@pytest.mark.parametrize("gain", [1e-6, 1.0, 3e3])
def test_operator_fast_path_arithmetics_agree_at_any_filter_magnitude(gain):
    # round 5: fp32 tensors in the model's configuration contract on the fp16 matrix cores -- every fp32 product as three fp16 MFMAs on split
    # operands (dcn2_kernel<x3_t>), the filters scaled by a power of two the DEVICE derives from max |w| (pack kernels -> the word behind the
    # bias), so that tiny filters do not fall into fp16's subnormals and large ones do not overflow.  Against the fp64 oracle and against
    # the exact fp32-MFMA arithmetic (dcn_v2.OP_F32_MFMA / H3D_DCN_F32_MFMA), both forms of the operator (packed + cached, and _ws)
    import ctypes
    B, C, Co, H, W = 2, 64, 96, 24, 40
    x = rnd("x", (B, C, H, W), -2.0, 2.0)
    w = rnd("w", (Co, C, 3, 3)) * (1.5 / np.sqrt(C * 9)) * gain
    b = rnd("b", (Co,)) * gain
    off = rnd("off", (B, 18, H, W), -3.0, 3.0)
    m = rnd("m", (B, 9, H, W), 0.0, 1.0)
    ref = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
    scale = float(ref.abs().max())
    y3 = _fwd(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    dcn_v2.OP_F32_MFMA = True
    try:
        y32 = _fwd(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    finally:
        dcn_v2.OP_F32_MFMA = False
    e3, e32 = float((y3 - ref).abs().max()) / scale, float((y32 - ref).abs().max()) / scale
    print("gain %g: relative error vs fp64 oracle: f16x3 %.3g, fp32 MFMA %.3g" % (gain, e3, e32))
    assert e3 <= 3e-6 and e32 <= 3e-6, (gain, e3, e32)
    assert not torch.equal(y3, y32)                      # (they ARE different kernels)
    # the reference-contract form (per-call pack inside the workspace)
    xd, wd, bd, od, md = [t.contiguous().to(DEV) for t in (x, w, b, off, m)]
    L = _lib.lib()
    nws = int(L.h3d_dcn_v2_workspace_bytes(B, C, H, W, Co))
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    out = torch.empty(B, Co, H, W, device=DEV)
    _lib.check(L.h3d_dcn_v2_forward_ws(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(od), _lib.ptr(md), _lib.ptr(out), B, C, H, W, Co,
                                       3, 3, 1, 1, 1, 1, 1, 1, 1, _lib.ptr(ws), nws, _lib.stream_ptr()), "forward_ws")
    assert torch.equal(out.cpu(), y3)                    # same kernel, same pack, same maximum


@pytest.mark.parametrize("gain", [1e-5, 1.0, 1e3])
def test_dcn_module_fused_launch_arithmetics_agree(gain):
    # the stand-alone `DCN` module in the model's configuration: ONE fused launch (offset conv + sampling + contraction); since round 5 on the
    # fp16 matrix cores (split operands, the filters of both convolutions scaled from the maxima the pack kernel leaves on the device),
    # against the fp64 oracle and against the fp32-matrix-instruction arithmetic (dcn_v2.OP_F32_MFMA)
    torch.manual_seed(7)
    dcn = dcn_v2.DCN(64, 96, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV).eval()
    with torch.no_grad():
        dcn.weight.mul_(gain)
        dcn.bias.copy_(torch.randn(96, device=DEV) * gain)
        dcn.conv_offset_mask.weight.copy_(torch.randn_like(dcn.conv_offset_mask.weight) * 0.02)
        dcn.conv_offset_mask.bias.copy_(torch.randn_like(dcn.conv_offset_mask.bias) * 0.3)
    x = rnd("xm", (2, 64, 40, 72), -2.0, 2.0)
    y3 = dcn(x.to(DEV)).cpu()
    dcn_v2.OP_F32_MFMA = True
    try:
        y32 = dcn(x.to(DEV)).cpu()
    finally:
        dcn_v2.OP_F32_MFMA = False
    ref = odcn.dcn_module_forward(x, dcn.weight.detach().cpu(), dcn.bias.detach().cpu(), dcn.conv_offset_mask.weight.detach().cpu(),
                                  dcn.conv_offset_mask.bias.detach().cpu(), acc_dtype=torch.float64)
    scale = float(ref.abs().max())
    e3, e32 = float((y3 - ref).abs().max()) / scale, float((y32 - ref).abs().max()) / scale
    print("DCN module, gain %g: relative error vs fp64 oracle: f16x3 %.3g, fp32 MFMA %.3g" % (gain, e3, e32))
    assert e3 <= 2e-5 and e32 <= 2e-5, (gain, e3, e32)       # (offsets computed in fp32-grade arithmetic move the sampling positions by ~1e-6 px)
    assert not torch.equal(y3, y32)


def test_operator_cached_pack_follows_in_place_filter_edits_of_any_magnitude():
    # the kept pack (h3d_dcn_v2_pack_weights_cached) validates itself on the device; its filter maximum -- what the split-operand kernel
    # scales by -- must be rebuilt with it: shrink the SAME weight tensor by 1e-6 in place (a stale maximum would leave every filter in
    # fp16's subnormal range: 1e-3 relative), then grow it by 1e9 (a stale maximum would overflow fp16: inf / NaN)
    B, C, Co, H, W = 1, 32, 48, 20, 24
    x = rnd("x", (B, C, H, W), -2.0, 2.0).to(DEV)
    w = (rnd("w", (Co, C, 3, 3)) * 0.1).to(DEV)
    b = torch.zeros(Co, device=DEV)
    off = rnd("off", (B, 18, H, W), -2.0, 2.0).to(DEV)
    m = rnd("m", (B, 9, H, W), 0.0, 1.0).to(DEV)
    for factor in (1.0, 1e-6, 1e9):
        w.mul_(factor)                                   # in place: same data_ptr, no torch version bump through .data
        with torch.no_grad():
            y = dcn_v2.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
        ref = odcn.dcn_v2_forward(x.cpu(), w.cpu(), b.cpu(), off.cpu(), m.cpu(), 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
        assert bool(torch.isfinite(y).all()), factor
        e = float((y - ref).abs().max()) / float(ref.abs().max())
        assert e <= 3e-6, (factor, e)


def test_operator_fast_path_random_shapes_vs_general_kernel():
    # 24 seeded random shapes of the model's configuration (any batch, C a multiple of 16, ANY Cout, maps from 1 x 1 to 40 x 40: partial
    # tiles, maps smaller than the apron, a single pixel) through the LDS-apron + split-operand MFMA kernel against the general
    # operator kernel (fp32 vector FMAs, the simple restatement of dcn_v2_im2col_cuda.cu) on the same operands
    rng = np.random.RandomState(2026)
    L = _lib.lib()
    for it in range(24):
        B, C = int(rng.randint(1, 4)), 16 * int(rng.randint(1, 6))
        Co = int(rng.choice([1, 7, 27, 33, 64, 100, 130]))
        H, W = int(rng.randint(1, 41)), int(rng.randint(1, 41))
        x = rnd("x%d" % it, (B, C, H, W), -2.0, 2.0)
        w = rnd("w%d" % it, (Co, C, 3, 3)) * (1.5 / np.sqrt(C * 9))
        b = rnd("b%d" % it, (Co,))
        off = rnd("o%d" % it, (B, 18, H, W), -1.0, 1.0) * float(rng.choice([0.5, 2.0, 8.0]))
        m = rnd("m%d" % it, (B, 9, H, W), 0.0, 1.0)
        y = _fwd(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        xd, wd, bd, od, md = [t.contiguous().to(DEV) for t in (x, w, b, off, m)]
        gen = torch.empty(B, Co, H, W, device=DEV)
        _lib.check(L.h3d_dcn_v2_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(od), _lib.ptr(md), _lib.ptr(gen), B, C, H, W,
                                        Co, 3, 3, 1, 1, 1, 1, 1, 1, 1, _lib.stream_ptr()), "general")
        e = float((y - gen.cpu()).abs().max())
        assert e <= 2e-5 * max(1.0, float(gen.abs().max())), (it, (B, C, Co, H, W), e)


def test_operator_boundary_gate_and_far_offsets():
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0
    off = torch.zeros(1, 18, 4, 4)
    off[0, 8, 0, :] = -0.5
    y = _fwd(x, w, torch.zeros(1), off, torch.ones(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    np.testing.assert_allclose(y[0, 0, 0].numpy(), np.full(4, 0.5, np.float32), atol=1e-7)
    off[0, 8, 0, :] = -1.0
    y = _fwd(x, w, torch.zeros(1), off, torch.ones(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    np.testing.assert_allclose(y[0, 0, 0].numpy(), np.zeros(4, np.float32), atol=1e-7)
    b = rnd("b", (3,))
    y = _fwd(rnd("x", (1, 2, 5, 5)), rnd("w", (3, 2, 3, 3)), b, torch.full((1, 18, 5, 5), 100.0), torch.ones(1, 9, 5, 5),
             3, 3, 1, 1, 1, 1, 1, 1, 1)
    np.testing.assert_allclose(y.numpy(), b.view(1, 3, 1, 1).expand(1, 3, 5, 5).numpy(), atol=1e-7)


def test_operator_errors_match_reference_messages():
    x = rnd("x", (1, 4, 8, 8)).to(DEV)
    w = rnd("w", (4, 4, 3, 3)).to(DEV)
    b = torch.zeros(4, device=DEV)
    off = torch.zeros(1, 18, 8, 8, device=DEV)
    m = torch.ones(1, 9, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match="kernel shape wont match"):
        dcn_v2.dcn_v2_forward(x, w, b, off, m, 5, 5, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="kernel channels wont match"):
        dcn_v2.dcn_v2_forward(x, rnd("w2", (4, 3, 3, 3)).to(DEV), b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        dcn_v2.dcn_v2_forward(x.cpu(), w.cpu(), b.cpu(), off.cpu(), m.cpu(), 3, 3, 1, 1, 1, 1, 1, 1, 1)


def test_dcn_module_fresh_is_half_conv():
    torch.manual_seed(0)
    mod = dcn_v2.DCN(16, 8, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV).eval()
    x = rnd("x", (2, 16, 12, 12)).to(DEV)
    with torch.no_grad():
        y = mod(x).cpu()
        ref = (0.5 * F.conv2d(x, mod.weight, None, 1, 1) + mod.bias.view(1, -1, 1, 1)).cpu()
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert set(mod.state_dict()) == {"weight", "bias", "conv_offset_mask.weight", "conv_offset_mask.bias"}


def test_dcn_module_model_configuration_is_one_fused_launch_and_matches_oracle():
    # DCN(chi, cho, (3,3), 1, 1, 1, 1) as model.py:355 builds it, with a trained-like conv_offset_mask: the stand-alone module
    # runs the fused DeformConv kernel in fp32 (no torch conv), checked against DCN.forward restated (oracle.dcn_module_forward)
    torch.manual_seed(0)
    mod = dcn_v2.DCN(64, 48, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV).eval()
    with torch.no_grad():
        mod.conv_offset_mask.weight.copy_(rnd("ow", (27, 64, 3, 3)).to(DEV) * 0.08)
        mod.conv_offset_mask.bias.copy_(rnd("ob", (27,)).to(DEV) * 0.3)
        mod.bias.copy_(rnd("b", (48,)).to(DEV))
    x = rnd("x", (2, 64, 20, 28))
    with torch.no_grad():
        y = mod(x.to(DEV)).cpu()
        ref = odcn.dcn_module_forward(x, mod.weight.cpu(), mod.bias.cpu(), mod.conv_offset_mask.weight.cpu(),
                                      mod.conv_offset_mask.bias.cpu(), acc_dtype=torch.float64)
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=0, atol=2e-4)
        # a parameter update invalidates the cached pack
        mod.bias.add_(1.0)
        y2 = mod(x.to(DEV)).cpu()
    np.testing.assert_allclose(y2.numpy(), (ref + 1.0).numpy(), rtol=0, atol=2e-4)
    assert mod._fused_ok(x.to(DEV)) and not dcn_v2.DCN(16, 8, (3, 3), 2, 1).to(DEV)._fused_ok(x.to(DEV))


def test_operator_throughput_form_cached_weights_channels_last_and_bf16():
    """The drop-in operator without the per-call work of the reference contract (h3d_dcn_v2_forward_packed): packed filters
    cached per parameter version, channels-last input read in place, bf16 input on the network's bf16 DeformConv path.  Every
    form against the oracle; the fp32 forms bit-identical to each other (same kernel, same operands)."""
    from h3d_amd import dcn_v2 as dv
    torch.manual_seed(0)
    B, C, Co, H, W = 2, 64, 48, 20, 28
    x = rnd("x", (B, C, H, W))
    w = rnd("w", (Co, C, 3, 3)) * (1.5 / np.sqrt(C * 9))
    b = rnd("b", (Co,))
    off = rnd("off", (B, 18, H, W), -3.0, 3.0)
    m = torch.sigmoid(rnd("m", (B, 9, H, W), -2.0, 2.0))
    ref = odcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
    xd, wd, bd, od, md = [t.to(DEV) for t in (x, w, b, off, m)]
    dv._PACKED.clear()
    with torch.no_grad():
        y0 = dv.dcn_v2_forward(xd, wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        assert len(dv._PACKED) == 1
        y1 = dv.dcn_v2_forward(xd, wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)          # second call: the cached pack
        assert len(dv._PACKED) == 1 and torch.equal(y0, y1)
        np.testing.assert_allclose(y0.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4)
        # channels-last input: no relayout, channels-last output, same numbers
        xcl = xd.contiguous(memory_format=torch.channels_last)
        y2 = dv.dcn_v2_forward(xcl, wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        assert y2.is_contiguous(memory_format=torch.channels_last) and y2.shape == y0.shape and torch.equal(y2, y0)
        # a parameter update is seen by the pack's device-side validation (same buffer, re-packed) ...
        wd.mul_(2.0)
        y3 = dv.dcn_v2_forward(xd, wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        assert len(dv._PACKED) == 1
        ref3 = odcn.dcn_v2_forward(x, 2.0 * w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
        np.testing.assert_allclose(y3.cpu().numpy(), ref3.numpy(), rtol=0, atol=4e-4)
        # ... and so is an edit through `.data`, which does NOT bump tensor._version (the reference edits its parameters
        # exactly so: dcn_v2.py:80-81, DCNv2/test.py:21 `weight.data.zero_()`)
        v = wd._version
        wd.data.mul_(0.5)
        bd.data.add_(1.0)
        assert wd._version == v
        y4 = dv.dcn_v2_forward(xd, wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        np.testing.assert_allclose(y4.cpu().numpy(), (ref + 1.0).numpy(), rtol=0, atol=2e-4)
        # a second stream right behind the first call of a NEW layer: it validates (and packs) for itself before it reads
        w2 = (wd * 3.0).contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ya = dv.dcn_v2_forward(xd, w2, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        with torch.cuda.stream(side):
            yb_ = dv.dcn_v2_forward(xd, w2, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb_)
        # the cache holds no reference to the parameters
        import gc
        import weakref
        r = weakref.ref(w2)
        del w2, ya, yb_
        gc.collect()
        assert r() is None
        wd.data.mul_(2.0)                                     # (back to 2 w and b for the bf16 form below)
        bd.data.sub_(1.0)
        # bf16 channels-last input: the network's bf16 path (fp16 filters and blend), bf16 channels-last output
        xb = bf16_round(x)
        wh = (2.0 * w).half().float()
        refb = odcn.dcn_v2_forward(xb, wh, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)
        yb = dv.dcn_v2_forward(xb.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last), wd, bd, od, md,
                               3, 3, 1, 1, 1, 1, 1, 1, 1)
        assert yb.dtype == torch.bfloat16 and yb.is_contiguous(memory_format=torch.channels_last)
        scale = max(1.0, float(refb.abs().max()))
        assert float((yb.float().cpu() - refb).abs().max()) <= 1.2e-2 * scale
        with pytest.raises(RuntimeError, match="channels_last"):
            dv.dcn_v2_forward(xd.to(torch.bfloat16), wd, bd, od, md, 3, 3, 1, 1, 1, 1, 1, 1, 1)


def test_dcn_module_runs_outside_no_grad_like_the_reference_module():
    # nn.Parameter requires grad by default, so `model.eval(); dcn(x)` outside torch.no_grad() used to raise.
    # The reference module runs there; ours computes without a graph and returns a detached tensor.  An input that itself
    # requires grad asks for dcn_v2_backward (out of scope): that still raises.
    torch.manual_seed(0)
    mod = dcn_v2.DCN(32, 16, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV).eval()
    x = rnd("x", (1, 32, 12, 12)).to(DEV)
    y = mod(x)
    assert not y.requires_grad
    with torch.no_grad():
        np.testing.assert_array_equal(y.cpu().numpy(), mod(x).cpu().numpy())
    with pytest.raises(RuntimeError, match="inference-only"):
        mod(x.clone().requires_grad_(True))
    # the general (non-fused) configuration goes through the operator: same rule
    mod2 = dcn_v2.DCN(16, 8, kernel_size=(3, 3), stride=2, padding=1).to(DEV).eval()
    assert not mod2(rnd("x2", (1, 16, 12, 12)).to(DEV)).requires_grad
    # training mode with grad enabled (a freshly constructed module: the reference's plain usage `DCN(...).cuda()(x)`, DCNv2/test.py:169-180):
    # the forward result comes back; a fine-tuning loop must not silently get no gradients: its backward raises
    yt = mod.train()(x)
    assert torch.equal(yt.detach(), y) and yt.requires_grad
    with pytest.raises(RuntimeError, match="inference-only"):
        yt.sum().backward()
    with torch.no_grad():
        assert torch.equal(mod.train()(x), y)
    mod.eval()


@pytest.mark.parametrize("cfg", [dict(dg=2), dict(stride=2), dict(dg=4, cin=48, cout=20), dict(k=1, padding=0), dict(dilation=2, padding=1)])
def test_dcn_module_outside_the_model_configuration_vs_oracle_without_any_vendor_conv(cfg, monkeypatch):
    """The reference's own example `DCN(64, 64, 3x3, dg=2)` on [2,64,128,128] (DCNv2/test.py:169-180) and other module
    configurations the fused launch does not cover: conv_offset_mask -> chunk / cat / sigmoid runs on this library's general
    kernel (`h3d_dcn_offset_mask`), then the operator -- nn.Conv2d.forward and F.conv2d are patched to raise while the product
    runs (round 4 called nn.Conv2d, i.e. MIOpen, here)."""
    torch.manual_seed(0)
    k, dg, cin, cout = cfg.get("k", 3), cfg.get("dg", 1), cfg.get("cin", 64), cfg.get("cout", 64)
    stride, padding, dilation = cfg.get("stride", 1), cfg.get("padding", 1), cfg.get("dilation", 1)
    mod = dcn_v2.DCN(cin, cout, kernel_size=(k, k), stride=stride, padding=padding, dilation=dilation, deformable_groups=dg).to(DEV)
    if dilation != 1:
        # the reference's conv_offset_mask takes no dilation (dcn_v2.py:107-111): its output only matches the operator's grid when
        # (H + 2p - k) / s == (H + 2p - d(k-1) - 1) / s, which no 3x3 d=2 module satisfies -- the operator's shape check must say so
        with pytest.raises(RuntimeError, match="offset shape"):
            mod(rnd("x", (1, cin, 12, 12)).to(DEV))
        return
    with torch.no_grad():
        mod.conv_offset_mask.weight.copy_(rnd("ow", tuple(mod.conv_offset_mask.weight.shape)).to(DEV) * 0.08)
        mod.conv_offset_mask.bias.copy_(rnd("ob", tuple(mod.conv_offset_mask.bias.shape)).to(DEV) * 0.3)
        mod.bias.copy_(rnd("b", (cout,)).to(DEV))
    shape = (2, 64, 128, 128) if cfg == dict(dg=2) else (2, cin, 21, 26)
    x = rnd("x", shape)
    assert not mod._fused_ok(x.to(DEV))
    sd = {n: v.detach().cpu() for n, v in mod.state_dict().items()}
    ref = odcn.dcn_module_forward(x, sd["weight"], sd["bias"], sd["conv_offset_mask.weight"], sd["conv_offset_mask.bias"], stride=stride,
                                  padding=padding, dilation=dilation, dg=dg, acc_dtype=torch.float64)

    def boom(*a, **kw):
        raise AssertionError("a torch convolution was called inside h3d_amd.dcn_v2.DCN.forward")
    monkeypatch.setattr(torch.nn.Conv2d, "forward", boom)
    monkeypatch.setattr(torch.nn.functional, "conv2d", boom)
    y = mod(x.to(DEV))           # (training mode, grad enabled: the reference's plain usage)
    monkeypatch.undo()
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=0, atol=3e-4)


def test_dcn_offset_mask_entry_point_vs_torch():
    """`h3d_dcn_offset_mask` alone: offset = the first 2k channels of Conv2d(x), mask = sigmoid of the last k (dcn_v2.py:119-124), ragged
    sizes, stride 2, a 5x3 kernel."""
    for (C, kh, kw, sh, sw, ph, pw, dg, H, W) in [(16, 3, 3, 1, 1, 1, 1, 1, 9, 13), (7, 5, 3, 2, 1, 2, 0, 2, 17, 11), (32, 1, 1, 1, 1, 0, 0, 3, 6, 6)]:
        k = dg * kh * kw
        x, w, b = rnd("x", (2, C, H, W)), rnd("w", (3 * k, C, kh, kw), -0.2, 0.2), rnd("b", (3 * k,))
        ref = F.conv2d(x.double(), w.double(), b.double(), (sh, sw), (ph, pw))
        Ho, Wo = ref.shape[2:]
        off = torch.empty(2, 2 * k, Ho, Wo, device=DEV)
        m = torch.empty(2, k, Ho, Wo, device=DEV)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)          # (kept alive across the launch)
        _lib.check(_lib.lib().h3d_dcn_offset_mask(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(off), _lib.ptr(m), 2, C, H, W,
                                                  kh, kw, sh, sw, ph, pw, dg, _lib.stream_ptr()), "dcn_offset_mask")
        np.testing.assert_allclose(off.cpu().numpy(), ref[:, :2 * k].float().numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(m.cpu().numpy(), torch.sigmoid(ref[:, 2 * k:]).float().numpy(), rtol=0, atol=2e-6)


def test_dcn_module_and_operator_on_a_second_stream_do_not_touch_the_owner_streams_pack():
    """the kept packs are validated through ONE 16-byte accumulator, so they belong to the stream that created them; a call
    on any other stream packs into buffers of its own.  Interleaved forwards on two streams (with a parameter edit in between) must both
    see the current parameters and agree bit for bit."""
    torch.manual_seed(3)
    mod = dcn_v2.DCN(32, 24, kernel_size=(3, 3), stride=1, padding=1).to(DEV).eval()
    with torch.no_grad():
        mod.conv_offset_mask.weight.copy_(rnd("ow", (27, 32, 3, 3)) * 0.05)
    x = rnd("x", (2, 32, 14, 18)).to(DEV)
    off, m = (rnd("off", (2, 18, 14, 18), -2, 2).to(DEV), rnd("m", (2, 9, 14, 18), 0, 1).to(DEV))
    y0 = mod(x)
    z0 = dcn_v2.dcn_v2_conv(x, off, m, mod.weight, mod.bias, 1, 1, 1, 1)
    owner = mod._pack[0].data_ptr()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y1 = mod(x)
        z1 = dcn_v2.dcn_v2_conv(x, off, m, mod.weight, mod.bias, 1, 1, 1, 1)
        mod.weight.data.mul_(2.0)
        y2 = mod(x)
        z2 = dcn_v2.dcn_v2_conv(x, off, m, mod.weight, mod.bias, 1, 1, 1, 1)
    side.synchronize()
    assert mod._pack[0].data_ptr() == owner            # the kept pack still belongs to the first stream
    assert torch.equal(y0, y1) and torch.equal(z0, z1)
    y3 = mod(x)
    z3 = dcn_v2.dcn_v2_conv(x, off, m, mod.weight, mod.bias, 1, 1, 1, 1)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3) and torch.equal(z2, z3) and not torch.equal(y0, y3)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_dcn_module_on_a_non_current_device():
    """the pack kernels must run on the input's device and stream, not on the current device's."""
    torch.manual_seed(4)
    mod = dcn_v2.DCN(32, 24, kernel_size=(3, 3), stride=1, padding=1).to("cuda:1").eval()
    x = rnd("x", (1, 32, 10, 10))
    with torch.cuda.device(0):
        y = mod(x.to("cuda:1"))
    ref = mod.to(DEV)(x.to(DEV))
    assert torch.equal(y.cpu(), ref.cpu())


def test_dcn_module_sees_parameter_edits_through_data():
    # the module keeps its packed filters and validates them on the device at every forward: `.data` edits (no version bump;
    # DCNv2/test.py:21, dcn_v2.py:80-81,114-116), load_state_dict and a move to the device all show up in the next forward
    torch.manual_seed(1)
    mod = dcn_v2.DCN(32, 24, kernel_size=(3, 3), stride=1, padding=1).to(DEV).eval()
    x = rnd("x", (2, 32, 14, 18))
    with torch.no_grad():
        mod.conv_offset_mask.weight.copy_(rnd("ow", (27, 32, 3, 3)) * 0.05)
        mod.conv_offset_mask.bias.copy_(rnd("ob", (27,)) * 0.5)

    def oracle():
        sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        om = F.conv2d(x, sd["conv_offset_mask.weight"], sd["conv_offset_mask.bias"], padding=1)
        o1, o2, mk_ = torch.chunk(om, 3, dim=1)
        return odcn.dcn_v2_forward(x, sd["weight"], sd["bias"], torch.cat((o1, o2), 1).contiguous(), torch.sigmoid(mk_).contiguous(),
                                   3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64)

    y0 = mod(x.to(DEV))
    np.testing.assert_allclose(y0.cpu().numpy(), oracle().numpy(), rtol=0, atol=3e-4)
    v = mod.weight._version
    mod.weight.data.mul_(-1.5)
    mod.bias.data.add_(0.25)
    mod.conv_offset_mask.weight.data.mul_(2.0)
    assert mod.weight._version == v
    y1 = mod(x.to(DEV))
    np.testing.assert_allclose(y1.cpu().numpy(), oracle().numpy(), rtol=0, atol=3e-4)
    assert float((y1 - y0).abs().max()) > 1e-2
    mod.conv_offset_mask.weight.data.zero_()
    mod.conv_offset_mask.bias.data.zero_()                   # init_offset (dcn_v2.py:114-116): fresh DCN == 0.5 * conv2d + b
    y2 = mod(x.to(DEV))
    ref2 = 0.5 * F.conv2d(x, mod.weight.detach().cpu(), None, padding=1) + mod.bias.detach().cpu().view(1, -1, 1, 1)
    np.testing.assert_allclose(y2.cpu().numpy(), ref2.numpy(), rtol=0, atol=3e-4)
    assert torch.equal(mod(x.to(DEV)), y2)                    # unchanged parameters: the kept pack, same bits


NET_CASES = [(2, 64, 64, 16, 16), (1, 128, 64, 24, 40), (1, 256, 128, 16, 16), (1, 512, 256, 8, 8), (1, 32, 16, 20, 20)]


@pytest.mark.parametrize("kind", ["dcn2", pytest.param("dcn_v1", marks=pytest.mark.extra)])      # (dcn_v1: csrc/dcn1.hip, `make EXTRA=1`)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", NET_CASES)
def test_network_dcn_op_vs_oracle(case, dtype, kind):
    B, Ci, Co, H, W = case
    x = rnd("x", (B, Ci, H, W))
    w = rnd("w", (Co, Ci, 3, 3)) * (1.5 / np.sqrt(Ci * 9))
    b = rnd("b", (Co,))
    om = rnd("om", (B, 27, H, W), -2.5, 2.5)          # raw conv_offset_mask output
    om[:, :18, :2, :] *= 4.0                              # some samples far outside the image
    half_w = (dtype == "bf16" and kind == "dcn2")            # generation-2 kernel: fp16 weights + fp16 blend
    if dtype == "bf16":
        x = bf16_round(x)
        w = w.half().float() if half_w else bf16_round(w)
    offset, mask = om[:, :18].contiguous(), torch.sigmoid(om[:, 18:]).contiguous()
    ref = F.relu(odcn.dcn_v2_forward(x, w, b, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, acc_dtype=torch.float64))
    xb, xp = nhwc(x, dtype)
    omb = torch.zeros(B, H, W, 32, dtype=torch.float32, device=DEV)
    omb[..., :27] = om.permute(0, 2, 3, 1).to(DEV)
    wp, bp, cout, rows = pack_conv(w, b, dtype)
    if half_w:
        wp = pack_conv(w, b, "f32")[0].half()
    out = torch.zeros(B, H, W, Co, dtype=TD[dtype], device=DEV)
    run(mk(_lib.OP_DCN if kind == "dcn2" else _lib.OP_DCN_V1, dtype, in_=xp, in2=omb.data_ptr(), w=wp.data_ptr(), bias=bp.data_ptr(), out=out.data_ptr(), B=B,
           H=H, W=W, Cin=Ci, in_cs=Ci, in2_cs=32, Ho=H, Wo=W, Cout=Co, out_cs=Co, ksize=3, stride=1, relu=1,
           out_mode=_lib.OUT_NHWC, wrows=rows))
    got = from_nhwc(out, Co)
    scale = max(1.0, float(ref.abs().max()))
    # bf16: the sampled operand is rounded to bf16 before the MFMA (2^-9 relative per term)
    tol = 5e-5 * scale if dtype == "f32" else 2.5e-2 * scale
    assert float((got - ref).abs().max()) <= tol
