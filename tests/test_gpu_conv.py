"""GPU parity: the MFMA conv / stem / pool / upsample-add / layout kernels against plain PyTorch
fp32 (CPU) of the same op.  Tolerances: f32 mode 2e-5 relative to the output scale (exact fmaf
chains, only summation order differs); bf16 mode inputs/weights are pre-rounded to bf16 so the
only error is fp32 accumulation order + the final bf16 rounding (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import DEV, TD, TN, bf16_round, conv, from_nhwc, lowp_round, mk, nhwc, rnd, run
from h3d_amd import _lib

pytestmark = pytest.mark.gpu


def _check(got, ref, dtype, what=""):
    scale = max(1.0, float(ref.abs().max()))
    # (fp16: the final rounding, 2^-11 relative; f16x3: fp32 storage, split-operand products -- the dropped lo.lo term is 2^-22
    #  relative per product and the lo terms of small values are fp16 subnormals, 3e-8 absolute: the f32 bound holds)
    tol = {"f32": 3e-5, "bf16": 1.2e-2, "f16": 1.5e-3, "f16x3": 3e-5}[dtype] * scale
    err = float((got - ref).abs().max())
    assert err <= tol, "%s: max err %.3g > %.3g" % (what, err, tol)


CONV_CASES = [
    # (B, Cin, Cout, H, W, k, stride, relu, residual)
    (2, 16, 16, 32, 32, 3, 1, True, False),     # level0 class (CK=16, MT=1)
    (1, 16, 32, 40, 24, 3, 2, True, False),     # level1 class, partial tiles
    (2, 32, 64, 24, 40, 3, 2, True, False),     # tree conv1 stride 2
    (2, 64, 64, 16, 16, 3, 1, True, True),      # BasicBlock conv2 + residual
    (1, 64, 64, 20, 36, 3, 1, False, False),    # partial tiles both ways
    (1, 128, 128, 16, 32, 3, 1, True, True),    # MT=4 path
    (1, 64, 256, 16, 16, 3, 1, True, False),    # head 3x3 (two cout chunks)
    (1, 256, 512, 8, 8, 3, 2, True, False),     # level5 conv1
    (2, 32, 64, 16, 16, 1, 1, False, False),    # project 1x1 (CK=16 path)
    (1, 448, 128, 16, 16, 1, 1, True, False),   # root 1x1 over a concat
    (1, 1280, 512, 4, 4, 1, 1, True, False),    # level5 root
    (2, 128, 64, 24, 24, 1, 1, True, False),    # level2 root (Cin % 64 == 0, <= 64 channels: 8-row tiles)
    (1, 32, 64, 40, 24, 3, 2, True, False),     # level2 tree1.conv1 (stride 2 below 64 input channels stays on conv.hip)
    (1, 128, 256, 24, 24, 3, 2, True, False),   # stride 2, 128-channel workgroups
    (1, 128, 192, 32, 40, 3, 1, True, True),    # f16x3: 32-row tiles (two N-tiles per wave, 64-channel blocks), partial tile in x
]


def _case_tensors(case, dtype):
    B, Ci, Co, H, W, k, s, relu, use_res = case
    x = rnd("x", (B, Ci, H, W))
    w = rnd("w", (Co, Ci, k, k), -1.0, 1.0) * (1.5 / np.sqrt(Ci * k * k))
    b = rnd("b", (Co,))
    x, w = lowp_round(x, dtype), lowp_round(w, dtype)
    return x, w, b


def conv_case_kernel_names(dtype):
    """Kernel instantiations CONV_CASES dispatch to (dry run; tests/test_gpu_variants.py's coverage check)."""
    names = set()
    for case in CONV_CASES:
        x, w, b = _case_tensors(case, dtype)
        res = rnd("r", (case[0], case[2], (case[3] - 1) // case[6] + 1, (case[4] - 1) // case[6] + 1)) if case[8] else None
        names.add(conv(x, w, b, dtype, stride=case[6], relu=case[7], res=res, name_only=True))
    return names


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch(case, dtype):
    B, Ci, Co, H, W, k, s, relu, use_res = case
    x, w, b = _case_tensors(case, dtype)
    ref = F.conv2d(x.double(), w.double(), b.double(), s, k // 2)
    res = None
    if use_res:
        res = lowp_round(rnd("r", tuple(ref.shape)), dtype)
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    got, _ = conv(x, w, b, dtype, stride=s, relu=relu, res=res)
    _check(got, ref.float(), dtype, str(case))


# 1x1 stride-1 bf16 convolutions as a GEMM (csrc/gemm1.hip).  reserved: 0x4000 = take the GEMM kernel whatever the grid,
# 0x100 / 0x200 / 0x300 = the 256x256 / 128x256 (three ring slots) / 128x128 (4 waves, two workgroups per CU) tile;
# 0x2000 = the halo-tile kernel instead.
GEMM1_CASES = [
    # (B, Cin, Cout, H, W, relu, residual, reserved, in_pad, out_pad[, stride])
    (2, 128, 64, 24, 24, True, False, 0x4300, 0, 0),       # DLA level2 root class (auto leaves <= 64 channels to conv.hip); 1152 px = 9 tiles
    (1, 448, 128, 16, 16, True, False, 0x4200, 0, 0),      # root over a concat, 7 stages through the 3-slot ring
    (1, 1280, 512, 8, 8, True, False, 0x4100, 0, 0),       # level5 root, 20 stages, 256x256 tile, 64 px: mostly empty tile
    (2, 256, 1024, 12, 12, False, True, 0x4100, 0, 0),     # ResNet bottleneck expand + residual, 288 px (partial tile)
    (2, 1024, 256, 12, 12, True, False, 0x4200, 0, 0),     # ResNet bottleneck reduce
    (1, 128, 72, 16, 16, False, False, 0x4200, 0, 0),      # Cout not a multiple of 32 (channel guard of the epilogue)
    (1, 192, 384, 16, 24, True, True, 0x4200, 0, 0),       # 384 = three channel blocks of 128
    (1, 192, 200, 16, 24, True, True, 0x4100, 0, 0),       # 200 of 256 packed rows: the rest of the 256-channel tile is zero filters
    (2, 128, 128, 16, 16, True, True, 0x4200, 64, 128),    # input / output inside wider concat buffers
    (1, 256, 64, 40, 40, True, False, 0x4300, 0, 64),      # 1600 px = 12.5 tiles of 128
    (4, 512, 256, 96, 96, True, False, 0, 0, 0),           # auto dispatch (grid large enough): 256x256
    (4, 512, 128, 96, 96, True, True, 0, 0, 0),            # auto: 128x128
    (8, 1024, 256, 48, 48, True, False, 0, 0, 0),          # auto: 72 tiles of 256x256 would leave most CUs idle: 128x128
    (2, 256, 512, 24, 40, False, False, 0x4100, 0, 0, 2),  # stride 2 (ResNet / Hourglass down-sampling 1x1): strided row gather
    (1, 128, 128, 17, 31, True, False, 0x4300, 64, 0, 2),  # stride 2, odd input size (9 x 16 outputs), input inside a wider buffer
    (8, 256, 512, 96, 96, False, False, 0, 0, 0, 2),       # stride 2, auto dispatch
]


def _gemm1_tensors(case, dtype="bf16"):
    B, Ci, Co, H, W, relu, use_res, reserved, in_pad, out_pad = case[:10]
    st = case[10] if len(case) > 10 else 1
    x = lowp_round(rnd("x", (B, Ci, H, W)), dtype)
    w = lowp_round(rnd("w", (Co, Ci, 1, 1)) * (1.5 / np.sqrt(Ci)), dtype)
    b = rnd("b", (Co,))
    res = lowp_round(rnd("r", (B, Co, (H - 1) // st + 1, (W - 1) // st + 1)), dtype) if use_res else None
    return x, w, b, res


def gemm1_case_kernel_names():
    return {conv(*_gemm1_tensors(c)[:3], "bf16", relu=c[5], res=_gemm1_tensors(c)[3], in_pad=c[8], out_pad=c[9], reserved=c[7],
                 stride=c[10] if len(c) > 10 else 1, name_only=True) for c in GEMM1_CASES}


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", GEMM1_CASES)
def test_gemm1_matches_torch_and_halo_kernel(case, dtype):
    B, Ci, Co, H, W, relu, use_res, reserved, in_pad, out_pad = case[:10]
    st = case[10] if len(case) > 10 else 1
    if dtype == "f16" and B * H * W > 100000:
        pytest.skip("fp16: the forced-tile cases cover every instantiation")
    x, w, b, res = _gemm1_tensors(case, dtype)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=st)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    kw = dict(relu=relu, res=res, in_pad=in_pad, out_pad=out_pad, stride=st)
    assert conv(x, w, b, dtype, reserved=reserved, name_only=True, **kw).startswith("gemm1_kernel<%s, " % TN[dtype])
    got, untouched = conv(x, w, b, dtype, reserved=reserved, **kw)
    _check(got, ref.float(), dtype, str(case))
    assert untouched is None or untouched
    # the halo-tile kernel of csrc/conv.hip on the same operands: same MFMA instruction over K in the same order
    assert conv(x, w, b, dtype, reserved=0x2000, name_only=True, **kw).startswith("conv_kernel<%s, " % TN[dtype])
    old, _ = conv(x, w, b, dtype, reserved=0x2000, **kw)
    assert float((got - old).abs().max()) <= (2e-2 if dtype == "bf16" else 3e-3) * max(1.0, float(ref.abs().max()))


def test_gemm1_declines_what_it_cannot_take():
    x, w, b = bf16_round(rnd("x", (1, 128, 9, 9))), bf16_round(rnd("w", (64, 128, 1, 1)) * 0.1), rnd("b", (64,))
    assert conv(x, w, b, "bf16", reserved=0x4000, name_only=True).startswith("conv_kernel<")      # 81 pixels: not a multiple of 16
    x = bf16_round(rnd("x", (1, 96, 16, 16)))
    w = bf16_round(rnd("w", (64, 96, 1, 1)) * 0.1)
    assert conv(x, w, b, "bf16", reserved=0x4000, name_only=True).startswith("conv_kernel<")      # Cin % 64 != 0
    x, w = rnd("x", (1, 128, 16, 16)), rnd("w", (64, 128, 1, 1)) * 0.1
    assert conv(x, w, b, "f32", reserved=0x4000, name_only=True).startswith("conv_kernel<")       # fp32 plans
    assert conv(bf16_round(x), bf16_round(w), b, "bf16", name_only=True).startswith("conv_kernel<")   # <= 64 output channels
    w, b = bf16_round(rnd("w", (128, 128, 1, 1)) * 0.1), rnd("b", (128,))
    assert conv(bf16_round(x), w, b, "bf16", name_only=True).startswith("conv_kernel<")                  # 2 x 1 tiles: grid too small


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x3"])
def test_conv_channel_strided_views(dtype):
    # input read from / output written into slices of wider concat buffers; neighbours untouched
    x = rnd("x", (2, 64, 16, 32))
    w = rnd("w", (64, 64, 3, 3)) * 0.06
    b = rnd("b", (64,))
    if dtype == "bf16":
        x, w = bf16_round(x), bf16_round(w)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1)).float()
    got, untouched = conv(x, w, b, dtype, relu=True, in_pad=64, out_pad=128)
    _check(got, ref, dtype)
    assert untouched


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f16x3"])
def test_conv_output_modes(dtype):
    x = rnd("x", (2, 256, 16, 24))
    w = rnd("w", (34, 256, 1, 1)) * 0.1
    b = rnd("b", (34,))
    x, w = lowp_round(x, dtype), lowp_round(w, dtype)
    ref = F.conv2d(x.double(), w.double(), b.double()).float()
    got, _ = conv(x, w, b, dtype, out_mode=_lib.OUT_NCHW_F32)
    tol = 3e-5 if dtype in ("f32", "f16x3") else 2e-4           # fp32 output: no bf16 rounding at the end
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    x = rnd("x", (1, 64, 16, 16))
    w = rnd("w", (27, 64, 3, 3)) * 0.05
    b = rnd("b", (27,))
    x, w = lowp_round(x, dtype), lowp_round(w, dtype)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1).float()
    got, _ = conv(x, w, b, dtype, out_mode=_lib.OUT_NHWC_F32, pad_cout_to=32)
    assert float((got[:, :27] - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert float(got[:, 27:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f16x3"])
def test_stem(dtype):
    x = rnd("img", (2, 3, 40, 56))
    w = rnd("w", (16, 3, 7, 7)) * 0.1
    b = rnd("b", (16,))
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 3)).float()
    xi = x.contiguous().to(DEV)
    wd, bd = w.contiguous().to(DEV), b.to(DEV)
    wexp = 0
    if dtype != "f32":                                    # MFMA stems: [16][7][32] k = dx*4+c; bf16 / fp16 image and weights, or (f16x3) split fp32
        ref = F.relu(F.conv2d(lowp_round(x, dtype).double(), lowp_round(w, dtype).double(), b.double(), 1, 3)).float()
        wp = torch.zeros(16, 7, 8, 4)
        wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        if dtype == "f16x3":
            from h3d_amd import engine
            wexp = engine.x3_exp(wp)
            wd = engine.x3_split(wp.reshape(16, 7, 32) * 2.0 ** wexp).contiguous().to(DEV)
        else:
            wd = wp.reshape(16, 7, 32).to(TD[dtype]).contiguous().to(DEV)
    out = torch.zeros(2, 40, 56, 16, dtype=TD[dtype], device=DEV)
    run(mk(_lib.OP_STEM, dtype, in_=xi.data_ptr(), w=wd.data_ptr(), bias=bd.data_ptr(), out=out.data_ptr(), B=2, H=40,
           W=56, Cin=3, in_cs=3, Ho=40, Wo=56, Cout=16, out_cs=16, ksize=7, stride=1, relu=1, wexp=wexp))
    _check(from_nhwc(out, 16), ref, dtype)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 64, 48, 80, 0), (1, 128, 37, 51, 0), (2, 64, 64, 64, 32), (1, 80, 23, 20, 0)])
def test_stem_stride2_matches_torch(shape, dtype):
    """csrc/extra.hip stem_s2_kernel (H3D_OP_STEM with stride 2, bf16 plans: the 7x7 stems of ResNet-101-DCN / Hourglass-104)
    against F.conv2d on the bf16-rounded operands; odd sizes, partial tiles, channel counts of 1.25 and 2 blocks of 64,
    output inside a wider buffer."""
    B, Co, H, W, out_pad = shape
    x = rnd("img", (B, 3, H, W))
    w = rnd("w", (Co, 3, 7, 7)) * 0.1
    b = rnd("b", (Co,))
    ref = F.relu(F.conv2d(lowp_round(x, dtype).double(), lowp_round(w, dtype).double(), b.double(), 2, 3)).float()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wp = torch.zeros(Co, 7, 8, 4)
    wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wd = wp.reshape(Co, 7, 32).to(TD[dtype]).contiguous().to(DEV)
    xi, bd = x.contiguous().to(DEV), b.to(DEV)
    cs = Co + out_pad
    out = torch.full((B, Ho, Wo, cs), 7.0, dtype=TD[dtype], device=DEV)
    op = mk(_lib.OP_STEM, dtype, in_=xi.data_ptr(), w=wd.data_ptr(), bias=bd.data_ptr(), out=out.data_ptr(), B=B, H=H, W=W, Cin=3,
            in_cs=3, Ho=Ho, Wo=Wo, Cout=Co, out_cs=cs, ksize=7, stride=2, relu=1)
    from gpu_helpers import kernel_name
    assert kernel_name(op) == "stem_s2_kernel<%s>" % TN[dtype]
    run(op)
    _check(from_nhwc(out, Co), ref, dtype, str(shape))
    if out_pad:
        assert bool((out[..., Co:].float() == 7.0).all().item())


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f16x3"])
def test_maxpool_and_upadd_and_copy(dtype):
    x = rnd("x", (2, 32, 18, 22))
    x = lowp_round(x, dtype)
    xb, xp = nhwc(x, dtype, 48, 8)
    out = torch.zeros(2, 9, 11, 32, dtype=TD[dtype], device=DEV)
    run(mk(_lib.OP_MAXPOOL, dtype, in_=xp, out=out.data_ptr(), B=2, H=18, W=22, Cin=32, in_cs=48, Ho=9, Wo=11, Cout=32,
           out_cs=32, ksize=2, stride=2))
    assert torch.equal(from_nhwc(out, 32), F.max_pool2d(x, 2, 2))
    x = lowp_round(rnd("xo", (1, 16, 19, 23)), dtype)                   # odd map: the last row / column is dropped (nn.MaxPool2d(2) floors)
    xb, xp = nhwc(x, dtype)
    out = torch.zeros(1, 9, 11, 16, dtype=TD[dtype], device=DEV)
    run(mk(_lib.OP_MAXPOOL, dtype, in_=xp, out=out.data_ptr(), B=1, H=19, W=23, Cin=16, in_cs=16, Ho=9, Wo=11, Cout=16, out_cs=16, ksize=2, stride=2))
    assert torch.equal(from_nhwc(out, 16), F.max_pool2d(x, 2, 2))
    for f, (uh, uw) in ((2, (6, 10)), (4, (6, 10)), (2, (5, 7)), (4, (3, 1)), (8, (2, 3))):
        k = 2 * f
        C = 64
        x = rnd("u", (2, C, uh, uw))
        skip = rnd("s", (2, C, uh * f, uw * f))
        w = rnd("w", (C, 1, k, k), 0.0, 1.0)
        x, skip = lowp_round(x, dtype), lowp_round(skip, dtype)
        ref = (F.conv_transpose2d(x.double(), w.double(), None, stride=f, padding=f // 2, groups=C) + skip.double()).float()
        xb, xp = nhwc(x, dtype)
        sb, sp = nhwc(skip, dtype)
        wd = w.reshape(C, k * k).t().contiguous().to(DEV)
        out = torch.zeros(2, uh * f, uw * f, C, dtype=TD[dtype], device=DEV)
        run(mk(_lib.OP_UPADD, dtype, in_=xp, in2=sp, w=wd.data_ptr(), out=out.data_ptr(), B=2, H=uh, W=uw, Cin=C, in_cs=C,
               in2_cs=C, Ho=uh * f, Wo=uw * f, Cout=C, out_cs=C, ksize=k, stride=f))
        _check(from_nhwc(out, C), ref, dtype, "upadd f=%d %dx%d" % (f, uh, uw))
    x = lowp_round(rnd("c", (1, 16, 5, 7)), dtype)
    xb, xp = nhwc(x, dtype)
    out = torch.zeros(1, 5, 7, 32, dtype=TD[dtype], device=DEV)
    run(mk(_lib.OP_COPY, dtype, in_=xp, out=out.data_ptr() + 16 * out.element_size(), B=1, H=5, W=7, Cin=16, in_cs=16,
           Ho=5, Wo=7, Cout=16, out_cs=32, ksize=1, stride=1))
    assert torch.equal(from_nhwc(out, 16, 16), x)


def test_layout_round_trip():
    x = rnd("x", (2, 19, 7, 9)).to(DEV)
    from gpu_helpers import HD
    for dtype in ("f32", "bf16", "f16"):
        mid = torch.zeros(2, 7, 9, 24, dtype=TD[dtype], device=DEV)
        _lib.check(_lib.lib().h3d_nchw_f32_to_nhwc(_lib.ptr(x), _lib.ptr(mid), HD[dtype], 2, 19, 7, 9, 24, _lib.stream_ptr()), "to_nhwc")
        back = torch.zeros_like(x)
        _lib.check(_lib.lib().h3d_nhwc_to_nchw_f32(_lib.ptr(mid), HD[dtype], _lib.ptr(back), 2, 19, 7, 9, 24, _lib.stream_ptr()), "to_nchw")
        torch.cuda.synchronize()
        exp = x if dtype == "f32" else x.to(TD[dtype]).float()
        assert torch.equal(back, exp)


def test_error_codes_raise():
    x = rnd("x", (1, 24, 8, 8))                    # Cin not a multiple of 16
    w = rnd("w", (16, 24, 3, 3))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        conv(x, w, None, "f32")
    x = rnd("x", (1, 16, 8, 8))
    w = rnd("w", (16, 16, 5, 5))
    with pytest.raises(RuntimeError, match="not covered"):
        conv(x, w, None, "bf16")


@pytest.mark.parametrize("shape", [(2, 48, 64), (1, 45, 72), (1, 45, 70), (3, 16, 32), (1, 130, 34), (1, 131, 36)])
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f16x3"])
def test_fused_stem_op_matches_torch(dtype, shape):
    """H3D_OP_STEM3 on its own (csrc/stem3.hip; f16x3: csrc/stem3x.hip): base_layer 7x7 3->16 + BN + ReLU -> level0 3x3 16->16 + BN + ReLU ->
    level1 3x3 stride 2 16->32 + BN + ReLU (model.py:231-249) against the same chain in fp64 on the CPU, on shapes the network never
    hands it: odd heights / widths, ragged 8 x 16 tiles, a single tile, a narrow tall image.  The 2-byte kernel reads the image as aligned
    float4 and must REFUSE a width that is not a multiple of 4 (this test found it computing the right edge from the next row's pixels
    instead: the launcher had no such check); the f16x3 kernel takes any width."""
    from gpu_helpers import fake_pw, mk, run
    B, H, W = shape
    sd = {}
    for name, (co, ci, k) in (("base.base_layer", (16, 3, 7)), ("base.level0", (16, 16, 3)), ("base.level1", (32, 16, 3))):
        sd[name + ".0.weight"] = rnd(name + "w", (co, ci, k, k)) * (1.8 / np.sqrt(ci * k * k))
        sd[name + ".1.weight"] = rnd(name + "g", (co,), 0.6, 1.4)
        sd[name + ".1.bias"] = rnd(name + "b", (co,), -0.3, 0.3)
        sd[name + ".1.running_mean"] = rnd(name + "m", (co,), -0.2, 0.2)
        sd[name + ".1.running_var"] = rnd(name + "v", (co,), 0.5, 1.5)
    pw = fake_pw(sd, dtype)
    x = rnd("img", (B, 3, H, W), -2.0, 2.0)
    wdev, bdev = pw.stem3_x3() if dtype == "f16x3" else pw.stem3()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    img = x.to(DEV).contiguous()
    out = torch.full((B, Ho, Wo, 32), float("nan"), dtype=TD[dtype], device=DEV)
    op = mk(_lib.OP_STEM3, dtype, in_=img.data_ptr(), w=wdev.data_ptr(), bias=bdev.data_ptr(), out=out.data_ptr(), B=B, H=H, W=W, Cin=3,
            in_cs=3, Ho=Ho, Wo=Wo, Cout=32, out_cs=32, ksize=7, stride=1, relu=1)
    if dtype != "f16x3" and W % 4:
        with pytest.raises(RuntimeError, match="multiple of 4"):
            run(op)
        return
    run(op)
    got = out.float().permute(0, 3, 1, 2).cpu()
    assert bool(torch.isfinite(got).all())
    t = x.double()
    for name, (k, s) in (("base.base_layer", (7, 1)), ("base.level0", (3, 1)), ("base.level1", (3, 2))):
        w, b = pw._fold(sd[name + ".0.weight"], None, name + ".1")
        if dtype != "f16x3":                    # the 2-byte plans round the image, the folded filters and both intermediates
            w, t = lowp_round(w, dtype), lowp_round(t.float(), dtype).double()
        t = F.relu(F.conv2d(t, w.double(), b.double(), s, k // 2))
    tol = {"bf16": 2e-2, "f16": 3e-3, "f16x3": 2e-5}[dtype] * max(1.0, float(t.abs().max()))
    assert float((got.double() - t).abs().max()) <= tol, (dtype, shape, float((got.double() - t).abs().max()), tol)


@pytest.mark.parametrize("shape", [(2, 13, 21), (1, 7, 5), (1, 33, 65), (3, 16, 32)])
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f16x3", "f32"])
def test_fused_heads_op_matches_torch(dtype, shape):
    """H3D_OP_HEADS on its own (csrc/heads.hip): per head Conv3x3(64 -> 256) + bias + ReLU -> Conv1x1(256 -> C) + bias (model.py:451-460) in
    one launch per width class, on map sizes no network produces (odd, smaller than a tile, one pixel past a tile) against fp64 on the CPU.
    The 256-channel intermediate is rounded to the plan's storage type on its way back into the matrix core."""
    import ctypes
    from gpu_helpers import fake_pw
    B, H, W = shape
    heads = {"hm": 1, "hps": 34, "pose": 72, "wh": 2}
    sd = {}
    for h, c in heads.items():
        sd[h + ".0.weight"] = rnd(h + "w1", (256, 64, 3, 3)) * (1.6 / np.sqrt(64 * 9))
        sd[h + ".0.bias"] = rnd(h + "b1", (256,), -0.2, 0.2)
        sd[h + ".2.weight"] = rnd(h + "w2", (c, 256, 1, 1)) * (1.6 / np.sqrt(256))
        sd[h + ".2.bias"] = rnd(h + "b2", (c,), -0.5, 0.5)
    pw = fake_pw(sd, dtype)
    pw.heads, pw.head_conv = dict(heads), 256
    x = lowp_round(rnd("feat", (B, 64, H, W), -1.5, 1.5), dtype)
    xb, xp = nhwc(x, dtype)
    outs, keep = {}, []
    for names in (("hm", "wh"), ("hps",), ("pose",)):                      # one launch per number of 32-row output tiles, as engine.Plan._lower_heads
        w1, b1, per = pw.fused_heads(names)
        desc = _lib.H3dHeadsDesc()
        desc.nheads = len(per)
        desc.wexp = pw.wexp.get(w1.data_ptr(), 0)
        for i, (hname, c, w2, b2) in enumerate(per):
            outs[hname] = torch.full((B, c, H, W), float("nan"), dtype=torch.float32, device=DEV)
            desc.head[i].w2, desc.head[i].b2, desc.head[i].out, desc.head[i].C = w2.data_ptr(), b2.data_ptr(), outs[hname].data_ptr(), c
            desc.head[i].wexp2 = pw.wexp.get(w2.data_ptr(), 0)
        keep.append(desc)
        run(mk(_lib.OP_HEADS, dtype, in_=xp, in2=ctypes.addressof(desc), w=w1.data_ptr(), bias=b1.data_ptr(), B=B, H=H, W=W, Cin=64, in_cs=64,
               Ho=H, Wo=W, Cout=256, ksize=3, stride=1))
    for h, c in heads.items():
        w1, w2 = lowp_round(sd[h + ".0.weight"], dtype), lowp_round(sd[h + ".2.weight"], dtype)
        t = F.relu(F.conv2d(x.double(), w1.double(), sd[h + ".0.bias"].double(), 1, 1))
        if dtype in ("bf16", "f16"):
            t = lowp_round(t.float(), dtype).double()
        ref = F.conv2d(t, w2.double(), sd[h + ".2.bias"].double())
        got = outs[h].cpu().double()
        assert bool(torch.isfinite(got).all()), h
        tol = {"bf16": 3e-2, "f16": 4e-3, "f16x3": 3e-5, "f32": 3e-5}[dtype] * max(1.0, float(ref.abs().max()))
        assert float((got - ref).abs().max()) <= tol, (dtype, shape, h, float((got - ref).abs().max()), tol)
