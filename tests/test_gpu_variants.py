"""GPU parity of EVERY tile configuration the LDS-DMA kernels can dispatch to, per op, through the C ABI
(h3d_run_ops): csrc/conv2.hip (`H3D_OP_CONV_STREAM`) against F.conv2d in fp64, csrc/dcn3.hip / dcn4.hip
(`H3D_OP_DCN_FUSED`, `_STREAM`, `_F16`, `H3D_OP_UPDCN_F16`) against the oracle (oracle/dcn.py, the restatement
of dcn_v2_im2col_cuda.cu:125-195 + dcn_v2.py:118-128).

Why this file exists: the launchers choose <MT, WAVES, S, SLOTS> from the workgroup count, so the small
network tests only ever reach the 4-wave variants while a batch-64 512x512 plan runs the 8/16-wave ones.
Here every variant is forced on small tensors with `h3d_op.reserved` and, separately, selected the natural
way by a tensor large enough; `test_bench_plan_kernels_are_all_covered` asserts that the set of kernel
instantiations tested here and in test_gpu_conv.py contains every instantiation of the batch-64 bench plan
and of the batch-32 plan (BASELINE configs 1 and 2).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_helpers import DEV, TN, Built, bf16_round, conv_stream_op, dcn_fused_op, dcn_fused_reference, kernel_name, lowp_round, rnd
from h3d_amd import _lib

pytestmark = pytest.mark.gpu

# ---- csrc/conv2.hip ------------------------------------------------------------------------------------
# (override, B, Cin, Cout, H, W, stride, relu, residual, in_pad, out_pad)
CONV2_CASES = [
    # forced variants on small, ragged tensors (H not a multiple of the tile height, W not of 16)
    (0x410, 2, 64, 128, 40, 24, 1, True, True, 0, 0),      # <4,16>: 32 x 16 px tiles, 16 waves
    (0x410, 1, 128, 256, 64, 32, 1, True, False, 0, 0),    #         two 128-channel groups, aligned
    (0x408, 2, 128, 128, 24, 40, 1, True, True, 0, 0),     # <4,8>
    (0x404, 1, 256, 256, 12, 20, 1, False, False, 0, 0),   # <4,4>
    (0x208, 2, 64, 64, 24, 40, 1, True, True, 64, 64),     # <2,8> inside concat buffers
    (0x204, 1, 32, 64, 20, 36, 1, True, False, 0, 0),      # <2,4>
    (0x108, 2, 32, 32, 24, 24, 1, True, False, 0, 0),      # <1,8>
    (0x104, 1, 48, 32, 10, 18, 1, False, False, 0, 0),     # <1,4>
    (0x5108, 2, 16, 16, 40, 40, 1, True, False, 0, 0),     # <1,8,1,1>: one stage, no ring
    (0x5208, 1, 64, 64, 24, 24, 1, True, True, 0, 0),      # <2,8,1,1>: one ring slot
    (0x5408, 1, 64, 128, 24, 24, 1, True, False, 0, 0),    # <4,8,1,1>
    (0x6410, 2, 64, 128, 40, 24, 1, True, True, 0, 0),     # <4,16,PIPE>: fragment reads one tap ahead, forced on a ragged tensor
    (0x6408, 1, 128, 128, 24, 40, 1, True, False, 0, 0),   # <4,8,PIPE>
    (0x6208, 2, 64, 64, 24, 40, 1, True, True, 64, 64),    # <2,8,PIPE> inside concat buffers
    (0x6404, 1, 96, 128, 12, 24, 1, False, False, 0, 0),   # <4,4,PIPE>
    (0x3208, 1, 96, 64, 24, 24, 1, True, False, 0, 0),     # three ring slots (counted vmcnt wait)
    (0x3108, 1, 96, 32, 24, 24, 1, True, False, 0, 0),
    (0x3404, 1, 96, 128, 12, 24, 1, True, False, 0, 0),
    (0x2408, 1, 64, 128, 40, 24, 1, True, True, 0, 0),     # two N-tiles per wave
    (0x2208, 1, 64, 64, 40, 24, 1, True, False, 0, 0),
    # stride 2 (33-pixel halo rows = two DMA pieces)
    (0, 2, 64, 128, 36, 40, 2, True, False, 0, 0),         # auto: <4,4,2,1>
    (0x2404, 1, 128, 256, 24, 24, 2, True, False, 0, 0),   # <4,4,2> two-slot ring
    (0x4408, 1, 64, 128, 40, 24, 2, True, False, 0, 0),    # <4,8,2,1>
    (0x4204, 1, 64, 64, 24, 40, 2, True, False, 0, 0),     # <2,4,2,1>
    (0, 1, 64, 64, 22, 34, 2, True, False, 0, 0),          # auto, odd sizes: <2,4,2>
    (0, 1, 64, 32, 24, 24, 2, False, False, 0, 0),         # auto: <1,4,2>
    # selected the natural way (>= 256 workgroups), the shapes of the batch-64 / batch-32 plans in miniature
    (0, 32, 32, 128, 64, 64, 1, True, True, 0, 0),         # level3/4 class: 256 tiles of 32 x 16 px -> <4,16>
    (0, 24, 32, 128, 48, 64, 1, True, True, 0, 0),         # level5 class: H not a multiple of 32 -> <4,8>
    (0, 4, 64, 64, 128, 128, 1, True, True, 0, 0),         # level2 class -> <2,8>
    # round 5: grids too small for 128-channel tiles (deep Hourglass levels, small DLA shards) take narrower channel blocks
    (0, 16, 128, 256, 4, 4, 1, True, False, 0, 0),         # 4 x 4 maps: <1,4>, 8 channel blocks per image
    (0, 16, 256, 256, 8, 8, 1, True, True, 0, 0),          # 8 x 8 maps: <1,4> (128 workgroups even at 32 channels)
    (0, 16, 64, 512, 4, 4, 1, True, False, 0, 0),          # 512 output channels: 32-channel blocks make 256 workgroups on 16-row tiles: <1,8>
    (0, 16, 128, 384, 32, 32, 1, True, True, 0, 0),        # 32 x 32 maps, 16 images: <2,8,PIPE>
    (0x10000000, 16, 128, 256, 4, 4, 1, True, False, 0, 0),  # ... round 4's rule on the same tensor: <4,4>
    # odd sizes (the C ABI takes any H, W)
    (0, 2, 128, 128, 13, 21, 1, True, True, 0, 0),
    (0, 1, 64, 64, 7, 5, 1, False, False, 0, 0),           # smaller than one tile both ways
    (0, 1, 64, 128, 11, 19, 2, True, False, 0, 0),         # stride 2, odd input: 6 x 10 outputs
    (0, 3, 128, 256, 5, 7, 2, True, True, 0, 0),           # stride 2, 3 x 4 outputs
]


def _conv2_built(case, dtype="bf16"):
    ov, B, Ci, Co, H, W, s, relu, use_res, ipad, opad = case
    x = lowp_round(rnd("x", (B, Ci, H, W)), dtype)
    w = lowp_round(rnd("w", (Co, Ci, 3, 3)) * (1.5 / np.sqrt(Ci * 9)), dtype)
    b = rnd("b", (Co,))
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = lowp_round(rnd("r", (B, Co, Ho, Wo)), dtype) if use_res else None
    return x, w, b, res, conv_stream_op(x, w, b, s, relu, res, ov, ipad, opad, dtype=dtype)


def test_conv2_small_grid_rule_selects_narrower_channel_blocks():
    # csrc/conv2.hip, round 5: below 256 workgroups at 128 channels per workgroup the launcher takes 64-channel 16-row tiles
    # if THEY reach 256, else 32-channel ones; 0x10000000 keeps round 4's 128-channel 8-row tile
    want = {(16, 128, 256, 4, 0): "conv2_kernel<unsigned short, 1, 4, ", (16, 256, 256, 8, 0): "conv2_kernel<unsigned short, 1, 4, ", (16, 64, 512, 4, 0): "conv2_kernel<unsigned short, 1, 8, ",
            (16, 128, 384, 32, 0): "conv2_kernel<unsigned short, 2, 8, ", (16, 128, 256, 4, 0x10000000): "conv2_kernel<unsigned short, 4, 4, "}
    seen = 0
    for c in CONV2_CASES:
        key = (c[1], c[2], c[3], c[4], c[0])
        if key in want and c[6] == 1:
            n = _conv2_built(c)[4].name
            assert n.startswith(want[key]), (c, n)
            seen += 1
    assert seen == len(want)


# fp16 plans (H3D_F16, BASELINE configs[4]) run the same templates with v_mfma_f32_32x32x16_f16 and an fp16 epilogue: the
# variants a ResNet-101-DCN / DLA-34 fp16 plan selects + one of every structural kind (PIPE, one slot, stride 2, residual,
# concat buffers)
CONV2_F16_CASES = [c for c in CONV2_CASES if c[0] in (0, 0x404, 0x6410, 0x6408, 0x6208, 0x5108, 0x4408, 0x204)]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", CONV2_CASES, ids=lambda c: "%#x-%dx%d-%dx%d-s%d-b%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[1]))
def test_conv_stream_variant_matches_torch(case, dtype):
    if dtype == "f16" and case not in CONV2_F16_CASES:
        pytest.skip("fp16: a subset of the variants")
    x, w, b, res, built = _conv2_built(case, dtype)
    assert built.name.startswith("conv2_kernel<%s, " % TN[dtype]), built.name
    s, relu = case[6], case[7]
    ref = F.conv2d(x.double(), w.double(), b.double(), s, 1)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    got = built.run()
    # bf16 / fp16 operands are exact in the reference, accumulation is fp32: what is left is the final rounding (2^-9 / 2^-12 relative)
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref.float()).abs().max())
    assert err <= (2.0 ** -8 if dtype == "bf16" else 2.0 ** -11) * scale, "%s: max err %.3g (scale %.2f)" % (built.name, err, scale)
    # the 16 x 32 px x 128 ch workgroups must be deterministic (DMA ring / counted waits)
    again = built.run()
    assert torch.equal(got, again), built.name


def test_conv_stream_auto_selection_reaches_the_wide_variants():
    names = {_conv2_built(c)[4].name for c in CONV2_CASES if c[0] == 0 and c[6] == 1}
    assert {"conv2_kernel<unsigned short, 4, 16, 2, 1, 2, 1, true>", "conv2_kernel<unsigned short, 4, 8, 2, 1, 2, 1, true>",
            "conv2_kernel<unsigned short, 2, 8, 2, 1, 2, 1, true>"} <= names, names


# ---- csrc/dcn3.hip, csrc/dcn4.hip ------------------------------------------------------------------------
# (kind, dtype, override, B, Cin, Cout, H, W, offset_scale)
DCN_CASES = [
    ("fused", "bf16", 0x400, 1, 128, 128, 24, 40, 0.5),    # dcn3<bf16,4,16,2>: every >= 128-channel layer of the batch-64 plan
    ("fused", "bf16", 0x400, 1, 256, 256, 16, 16, 3.0),
    ("fused", "bf16", 0x400, 1, 512, 256, 16, 16, 12.0),   #   ... with most samples through pass 2 (global gather)
    ("fused", "bf16", 0x200, 1, 256, 128, 24, 24, 0.5),    # dcn3<bf16,2,32,2> on a >64-channel layer (two channel groups)
    ("fused", "bf16", 0x200, 1, 144, 128, 16, 16, 3.0),    # dcn3<bf16,2,16,2> (Cin not a multiple of 32)
    ("fused", "bf16", 0, 2, 128, 64, 24, 40, 0.5),         # dcn3<bf16,2,32,2>
    ("fused", "bf16", 0, 1, 48, 64, 20, 20, 3.0),          # dcn3<bf16,2,16,2>
    ("fused", "bf16", 0, 1, 64, 32, 20, 36, 3.0),          # dcn3<bf16,1,32,2>
    ("fused", "bf16", 0, 1, 48, 32, 16, 16, 0.5),          # dcn3<bf16,1,16,2>
    ("fused", "f32", 0, 1, 64, 64, 20, 24, 3.0),           # parity mode
    ("fused", "f32", 0, 1, 32, 32, 16, 16, 12.0),
    ("fused", "f16x3", 0, 1, 64, 64, 20, 24, 3.0),         # parity arithmetic on the fp16 matrix cores: dcn3<x3_t,2,16,2>
    ("fused", "f16x3", 0, 2, 128, 64, 24, 40, 0.5),
    ("fused", "f16x3", 0, 1, 32, 32, 16, 16, 12.0),        # dcn3<x3_t,1,16,2>, most samples through pass 2
    ("fused", "f16x3", 0, 1, 256, 128, 16, 16, 3.0),       #   two channel groups
    ("fused", "f16x3", 0, 1, 64, 64, 64, 80, 3.0),         # dcn3<x3_t,2,16,6>: margin-6 apron on maps of >= 64 rows
    ("fused", "f16x3", 0x4000, 1, 32, 32, 16, 16, 12.0),   # dcn3<x3_t,1,16,6>
    ("fused", "f16x3", 0x2000, 1, 64, 64, 20, 24, 3.0),    # dcn3<x3_t,2,16,2>: the f32 plan's margin-2 double-buffered tile
    ("stream", "f16x3", 0, 2, 128, 64, 24, 40, 0.5),       # dcn3<x3_t,2,16,2,WDMA,256>: f16x3 with patch slots (fp32 list entries and patch units)
    ("stream", "f16x3", 0, 1, 256, 64, 16, 32, 3.0),       #   ... 4-13 % of the samples in patches, second fill round (entries >= 128)
    ("stream", "f16x3", 0, 2, 64, 64, 40, 24, 6.0),        #   ... more far samples than slots: patches AND pass 2
    ("stream", "f16x3", 0, 1, 64, 64, 16, 16, 40.0),       #   ... nearly every sample outside the apron or the image
    ("stream", "f16x3", 0, 1, 64, 32, 20, 20, 12.0),       # dcn3<x3_t,1,16,2,WDMA,256>
    ("stream", "f16x3", 0, 1, 256, 256, 16, 16, 3.0),      # dcn3<x3_t,2,16,4,WDMA,256>: margin 4 (a `node` layer above 64 channels), grid.y = 4
    ("stream", "f16x3", 0x8000, 1, 256, 256, 16, 16, 3.0), # dcn3<x3_t,2,16,3,WDMA,256>: margin 3
    ("stream", "f16x3", 0x10000, 1, 64, 32, 20, 20, 12.0), # dcn3<x3_t,1,16,4,WDMA,256>
    ("stream", "f16x3", 0x10000, 2, 64, 64, 40, 24, 6.0),  #   margin 4 with more far samples than slots: patches AND pass 2
    ("stream", "f16x3", 0x8000, 1, 64, 32, 20, 20, 12.0),  # dcn3<x3_t,1,16,3,WDMA,256>
    ("stream", "f16x3", 0x4000, 1, 256, 128, 16, 16, 6.0), # dcn3<x3_t,2,16,2,WDMA,256> forced on a wide layer
    ("stream", "f16x3", 0, 1, 64, 64, 13, 21, 3.0),        # odd map sizes (the C ABI takes any H, W)
    ("stream", "bf16", 0, 2, 64, 64, 13, 21, 3.0),
    ("stream", "bf16", 0, 1, 128, 128, 7, 5, 2.0),         # a map smaller than one tile both ways
    ("fused", "f32", 0, 1, 64, 64, 13, 21, 3.0),
    ("stream", "bf16", 0, 2, 128, 64, 24, 40, 0.5),        # dcn3<bf16,2,16,2,WDMA,256>: two workgroups per CU, patch slots
    ("stream", "bf16", 0, 1, 256, 64, 16, 32, 3.0),        #   ... 4-13 % of the samples in patches
    ("stream", "bf16", 0, 2, 64, 64, 40, 24, 6.0),         #   ... more samples leave the apron than a tile has slots: patches AND pass 2
    ("stream", "bf16", 0, 1, 64, 64, 16, 16, 40.0),        #   ... nearly every sample outside the apron or the image
    ("stream", "bf16", 0, 1, 64, 32, 20, 20, 12.0),        # dcn3<bf16,1,16,2,WDMA,256>
    ("stream", "bf16", 0x400, 1, 128, 128, 16, 32, 3.0),   # dcn3<bf16,4,16,4,WDMA,256>: margin-4 apron (0x400: also for a small grid)
    ("stream", "bf16", 0x400, 1, 256, 256, 24, 24, 8.0),
    ("stream", "bf16", 0, 1, 256, 256, 16, 16, 3.0),       # small grid: 64-channel workgroups on a 256-channel layer (grid.y = 4)
    ("stream", "bf16", 0, 12, 128, 128, 64, 64, 0.5),      # >= 192 workgroups: the 128-channel variant without an override
    ("stream", "bf16", 0, 1, 48, 64, 20, 20, 3.0),         # Cin = 16 (mod 32): falls back to the configuration without patches
    ("stream", "bf16", 0x1000, 2, 128, 64, 24, 40, 0.5),   # round 1's configurations: dcn3<bf16,2,16,1,WDMA,0>
    ("stream", "bf16", 0x1000, 1, 64, 32, 20, 20, 12.0),   # dcn3<bf16,1,16,1,WDMA,0>
    ("stream", "bf16", 0x1000, 1, 128, 128, 16, 32, 3.0),  # dcn3<bf16,4,16,2,WDMA,0>
    ("f16", "bf16", 0, 2, 64, 64, 24, 40, 0.5),            # dcn4<2,.,1,0> DENSE
    ("f16", "bf16", 0, 1, 64, 64, 20, 20, 12.0),
    ("f16", "bf16", 0x100, 1, 64, 64, 24, 40, 3.0),        # dcn4<2,.,0,0> one workgroup per CU
    ("f16", "bf16", 0, 1, 64, 32, 20, 36, 3.0),            # dcn4<1,.,1,0>
    ("f16", "bf16", 0x100, 1, 64, 32, 16, 16, 0.5),
    ("updcn2", "bf16", 0, 2, 64, 64, 12, 20, 0.5),         # dcn4<2,.,1,1>: 2x up-sampling + add folded in
    ("updcn2", "bf16", 0, 1, 64, 64, 10, 10, 12.0),
    ("updcn4", "bf16", 0, 1, 64, 64, 6, 10, 3.0),          # 4x (ida_up.up_2: 8x8 stride-4 deconv)
    ("updcn2", "bf16", 0, 1, 64, 32, 10, 18, 3.0),         # dcn4<1,.,1,1>
    # selected the natural way by the workgroup count (>= 192 x 128-channel workgroups / >= 512 tiles)
    ("fused", "bf16", 0, 12, 128, 128, 64, 64, 0.5),       # 192 workgroups -> MT = 4 without an override
    # fp16 plans (H3D_F16): the apron needs no conversion while it is staged; everything else as in bf16 plans
    ("stream", "f16", 0, 2, 128, 64, 24, 40, 0.5),         # dcn3<f16,2,16,2,WDMA,256>
    ("stream", "f16", 0, 1, 256, 64, 16, 32, 3.0),
    ("stream", "f16", 0, 2, 64, 64, 40, 24, 6.0),          #   ... patches AND pass 2
    ("stream", "f16", 0, 1, 64, 32, 20, 20, 12.0),         # dcn3<f16,1,16,2,WDMA,256>
    ("stream", "f16", 0x400, 1, 128, 128, 16, 32, 3.0),    # dcn3<f16,4,16,4,WDMA,256>: margin-4 apron
    ("stream", "f16", 0x400, 1, 256, 256, 24, 24, 8.0),    #   (ResNet-101-DCN's first up-sampling stage in miniature)
    # 0x8000: the wide-margin variants on the packed apron (margin 4 at two workgroups per CU; engine.dcn_wide_margin / calibration)
    ("stream", "bf16", 0x8000, 2, 128, 64, 24, 40, 0.5),   # dcn3<bf16,2,16,4,WDMA,256,PK>
    ("stream", "bf16", 0x8000, 1, 256, 64, 16, 32, 3.0),
    ("stream", "bf16", 0x8000, 2, 64, 64, 40, 24, 6.0),    #   ... patches AND pass 2
    ("stream", "bf16", 0x8000, 1, 64, 64, 16, 16, 40.0),
    ("stream", "bf16", 0x8000, 1, 64, 32, 20, 20, 12.0),   # dcn3<bf16,1,16,4,WDMA,256,PK>
    ("stream", "bf16", 0x8000, 1, 256, 256, 16, 16, 3.0),  # small grid: 64-channel workgroups, wide margin
    ("stream", "bf16", 0x8400, 1, 128, 128, 16, 32, 3.0),  # the 128-channel variant has margin 4 anyway
    ("stream", "f16", 0x8000, 2, 128, 64, 24, 40, 3.0),    # dcn3<f16,2,16,4,WDMA,256,PK>
    ("stream", "f16", 0x8000, 1, 64, 32, 20, 20, 12.0),
    # 0x10000: margin 2 on the packed apron with 512 patch slots, the second 256 filled in a second round per stage
    ("stream", "bf16", 0x10000, 2, 128, 64, 24, 40, 0.5),  # dcn3<bf16,2,16,2,WDMA,512,PK>: one round
    ("stream", "bf16", 0x10000, 2, 64, 64, 40, 24, 4.0),   #   ... tiles with 256-512 far samples: two rounds, no pass 2
    ("stream", "bf16", 0x10000, 2, 64, 64, 40, 24, 9.0),   #   ... more than 512: two rounds AND pass 2
    ("stream", "bf16", 0x10000, 1, 64, 32, 20, 20, 12.0),  # dcn3<bf16,1,16,2,WDMA,512,PK>
    ("stream", "bf16", 0x10400, 1, 256, 256, 24, 24, 8.0), # dcn3<bf16,4,16,4,WDMA,512>
    ("stream", "f16", 0x10000, 2, 64, 64, 40, 24, 4.0),
    # stream16: bf16 plan, fp16 INPUT (reserved | 0x40000, csrc/dcn3.hip F16IN: the `node` DeformConvs behind an up-sample + add that
    # writes fp16); bf16 output
    ("stream16", "bf16", 0, 2, 64, 64, 40, 24, 0.5),       # dcn3<bf16,2,16,2,WDMA,256,false,F16IN>: the five 64 -> 64 @128x128 nodes
    ("stream16", "bf16", 0, 2, 64, 64, 40, 24, 6.0),       #   ... patches AND pass 2
    ("stream16", "bf16", 0, 1, 64, 64, 16, 16, 40.0),
    ("stream16", "bf16", 0x400, 1, 128, 128, 16, 32, 3.0), # dcn3<bf16,4,16,4,WDMA,256,false,F16IN>: 128 -> 128 @64x64, 256 -> 256 @32x32
    ("stream16", "bf16", 0x400, 1, 256, 256, 24, 24, 8.0),
    ("stream16", "bf16", 0, 1, 256, 256, 16, 16, 3.0),     # small grid: 64-channel workgroups
    ("stream16", "bf16", 0x8000, 2, 64, 64, 40, 24, 6.0),  # wide margin, packed apron
    ("stream16", "bf16", 0x10000, 2, 64, 64, 40, 24, 4.0), # 512 slots, two rounds
    ("stream16", "bf16", 0x10400, 1, 256, 256, 24, 24, 8.0),
    # 0x4000: csrc/dcn5.hip (apron AND filters by LDS-DMA; measured slower, kept selectable: DESIGN.md 2.2)
    ("stream", "f16", 0x4000, 2, 128, 64, 24, 40, 0.5),    # dcn5<2,2,.,256>
    ("stream", "f16", 0x4000, 1, 256, 64, 16, 32, 3.0),
    ("stream", "f16", 0x4000, 2, 64, 64, 40, 24, 6.0),     #   ... patches AND pass 2
    ("stream", "f16", 0x4000, 1, 64, 32, 20, 20, 12.0),    # dcn5<1,2,.,256>
    ("stream", "f16", 0x4400, 1, 128, 128, 16, 32, 3.0),   # dcn5<4,4,.,256>
    ("stream", "f16", 0x4400, 1, 256, 256, 24, 24, 8.0),
    ("stream", "f16", 0x4000, 1, 48, 64, 20, 20, 3.0),     # three stages (no two-stage unrolling in dcn5)
    ("stream", "f16", 0x4000, 1, 64, 64, 16, 16, 40.0),
    ("stream", "f16", 0, 1, 256, 256, 16, 16, 3.0),        # small grid: 64-channel workgroups
    ("stream", "f16", 0, 1, 48, 64, 20, 20, 3.0),          # Cin = 16 (mod 32): no patches
    ("stream", "f16", 0, 1, 64, 64, 16, 16, 40.0),         #   ... nearly every sample outside the apron or the image
    ("stream", "f16", 0, 12, 128, 128, 64, 64, 0.5),       # >= 192 workgroups: the 128-channel variant without an override
    ("stream", "f16", 0x1000, 1, 64, 32, 20, 20, 12.0),    # 0x1000: no patch slots, dcn3<f16,1,16,1,WDMA,0>
    ("fused", "f16", 0, 1, 64, 32, 20, 36, 3.0),           # register-staged filters: dcn3<f16,1,32,2>
    ("fused", "f16", 0x400, 1, 128, 128, 24, 40, 0.5),     # dcn3<f16,4,16,2>
]


def _is_extra(case):
    """Cases of the superseded generations (csrc/dcn4.hip: kinds f16 / updcn*; csrc/dcn5.hip: fp16 stream cases with 0x4000): `make EXTRA=1`."""
    return case[0] in ("f16", "updcn2", "updcn4") or (case[0] == "stream" and case[1] == "f16" and bool(case[2] & 0x4000))


def _active_dcn_cases():
    from conftest import has_extra
    return [c for c in DCN_CASES if has_extra() or not _is_extra(c)]


def _dcn_built(case):
    kind, dtype, ov, B, Ci, Co, H, W, oscale = case
    x = rnd("x", (B, Ci, H, W))
    a = float(np.sqrt(3.0 / (Ci * 9)))
    w = rnd("w", (Co, Ci, 3, 3)) * (1.5 / np.sqrt(Ci * 9))
    b = rnd("b", (Co,))
    wo = rnd("wo", (27, Ci, 3, 3)) * (a * oscale * np.sqrt(3.0))      # offsets ~ N(0, oscale^2)-ish like synth weights
    bo = rnd("bo", (27,), -0.1, 0.1)
    skip = w_up = None
    if dtype in ("bf16", "f16"):
        w, wo = w.half().float(), wo.half().float()                       # DCN filters are fp16 in bf16 and fp16 plans
        x = lowp_round(x, dtype)
    if kind.startswith("updcn"):
        f = int(kind[-1])
        skip = bf16_round(rnd("skip", (B, Ci, H * f, W * f)))
        w_up = rnd("wup", (Ci, 1, 2 * f, 2 * f), 0.0, 0.5)
        xin = (F.conv_transpose2d(x.double(), w_up.double(), None, stride=f, padding=f // 2, groups=Ci)
               + skip.double()).float().half().float()                    # the folded sum is rounded once, to fp16
        built = dcn_fused_op("updcn", x, w, b, wo, bo, dtype, ov, skip, w_up)
    else:
        xin = x.half().float() if kind in ("f16", "stream16") else x
        built = dcn_fused_op(kind, xin if kind in ("f16", "stream16") else x, w, b, wo, bo, dtype, ov)
    return xin, w, b, wo, bo, built


def _dcn_id(c):
    return "%s-%s-%#x-%dx%d-%dx%d-o%g-b%d" % (c[0], c[1], c[2], c[4], c[5], c[6], c[7], c[8], c[3])


@pytest.mark.parametrize("case", [pytest.param(c, marks=pytest.mark.extra, id=_dcn_id(c)) if _is_extra(c) else pytest.param(c, id=_dcn_id(c)) for c in DCN_CASES])
def test_dcn_fused_variant_matches_oracle(case):
    kind, dtype = case[0], case[1]
    xin, w, b, wo, bo, built = _dcn_built(case)
    ref, om = dcn_fused_reference(xin, w, b, wo, bo)
    got = built.run()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    # f32: exact fmaf chains, offsets from an fp32 conv (sampling positions move by ~1e-6 px).
    # bf16: fp16 blend (2^-11 per sample) + f16 MFMA with fp32 accumulation + one bf16 rounding of the output (2^-9)
    # fp16: the same blend and MFMA, fp16 rounding of the output (2^-12)
    # f16x3: the fp32 blend and geometry, every product as three fp16 MFMAs on split operands: the f32 bound
    tol = 2e-4 * scale if dtype in ("f32", "f16x3") else (1.2e-2 if dtype == "bf16" else 4e-3) * scale
    frac_far = float((om[:, :18].abs() > 1.0).float().mean())
    assert err <= tol, "%s: max err %.3g > %.3g (scale %.2f, |offset|>1 for %.0f%%)" % (built.name, err, tol, scale, 100 * frac_far)
    assert torch.equal(got, built.run()), built.name


@pytest.mark.parametrize("dtype,ov", [("bf16", 0), ("f16", 0), pytest.param("f16", 0x4000, marks=pytest.mark.extra), ("bf16", 0x8000), ("bf16", 0x10000)])
def test_dcn_tiles_with_more_far_samples_than_patch_slots_are_deterministic(dtype, ov):
    # A tile with more than NP samples outside its apron sends the surplus through pass 2 (another accumulation order).  Round 2
    # handed out the slots with an LDS atomic per wave, so WHICH samples were the surplus depended on the order the waves arrived
    # in: last-bit differences from run to run (found at batch 8, 512 x 512, where the 256-channel 32 x 32 layer runs the
    # 64-channel-workgroup variant).  Slots are now assigned in (wave, tap, lane) order: many runs, four channel groups per
    # tile (grid.y = 4), most tiles over their slot count -- every run bit-identical.
    case = ("stream", dtype, ov, 4, 128, 256, 32, 32, 6.0)
    xin, w, b, wo, bo, built = _dcn_built(case)
    ref, om = dcn_fused_reference(xin, w, b, wo, bo)
    first = built.run()
    scale = max(1.0, float(ref.abs().max()))
    assert float((first - ref).abs().max()) <= (1.2e-2 if dtype == "bf16" else 4e-3) * scale
    for _ in range(8):
        assert torch.equal(first, built.run()), built.name


def test_dcn_bf16_input_beyond_the_fp16_range_is_clamped_not_overflowed():
    # bf16 plans sample from an fp16 apron: a FINITE bf16 activation beyond +-65504 SATURATES while it is staged (round toward
    # zero in v_cvt_pkrtz_f16_f32, SE<bf16_t>::convert16) -- in the LDS apron, in the patch pixels and in pass 2.
    # (+-inf / NaN activations are outside the contract: they pass through and 0 x inf = NaN can reach zero-weighted corners.)  Driven past the range here: the
    # result must equal the oracle's on the clipped input (no inf / nan), for samples inside and outside the apron.
    B, Ci, Co, H, W = 1, 64, 64, 24, 24
    x = bf16_round(rnd("x", (B, Ci, H, W)) * 1.0e5)                   # ~35 % of the values beyond 65504
    assert float((x.abs() > 65504).float().mean()) > 0.2
    a = float(np.sqrt(3.0 / (Ci * 9)))
    w = (rnd("w", (Co, Ci, 3, 3)) * (1.5 / np.sqrt(Ci * 9))).half().float()
    b = rnd("b", (Co,))
    wo = (rnd("wo", (27, Ci, 3, 3)) * (a * 6.0e-5)).half().float()      # offsets of a few pixels on inputs of 1e5
    bo = rnd("bo", (27,), -0.1, 0.1)
    for kind in ("stream", "fused"):
        built = dcn_fused_op(kind, x, w, b, wo, bo, "bf16", 0)
        got = built.run()
        assert bool(torch.isfinite(got).all()), built.name
        # the offset convolution sees the clamped input too (it reads the same apron)
        ref, om = dcn_fused_reference(x.clamp(-65504.0, 65504.0), w, b, wo, bo)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1.2e-2 * scale, (built.name, float((got - ref).abs().max()), scale)
        assert float((om[:, :18].abs() > 1.0).float().mean()) > 0.05      # (some samples do leave their pixel)


def test_dcn_wide_margin_variant_is_bit_identical_while_no_tile_overflows():
    # A sample inside the apron is blended from LDS in phase B; the same sample outside a narrower apron is blended by the same
    # fp16 chain when its patch pixel is filled and then read back with weights (1, 0, 0, 0): the same value.  So the margin-2 and the
    # margin-4 (packed apron) variants agree bit for bit as long as neither has a tile with more far samples than patch slots.
    # The same holds for the 512-slot variant (0x10000: packed margin-2 apron, second patch round) -- which is why the timed calibration
    # (DLAEngine.calibrate_dcn_margins), whose choices can differ from run to run, does not change a plan's outputs unless a tile
    # overflows its slots (then pass 2 and a patch accumulate in different orders: only the tolerance against the oracle is shared).
    for dtype in ("bf16", "f16"):
        for shape in ((2, 128, 64, 40, 56, 1.5), (1, 64, 64, 48, 48, 2.5), (1, 64, 32, 24, 40, 1.5)):
            narrow = _dcn_built(("stream", dtype, 0) + shape)[5]
            wide = _dcn_built(("stream", dtype, 0x8000) + shape)[5]
            assert ", 4, " in wide.name and wide.name.endswith(", true>") and not narrow.name.endswith(", true>"), (narrow.name, wide.name)
            ref = narrow.run().clone()
            assert torch.equal(ref, wide.run()), (dtype, shape)
            more = _dcn_built(("stream", dtype, 0x10000) + shape)[5]
            assert ", 512, true>" in more.name, more.name
            assert torch.equal(ref, more.run()), (dtype, shape, "512 slots")


def test_dcn_f16_stream_dispatch():
    from conftest import has_extra
    names = {c: _dcn_built(c)[5].name for c in _active_dcn_cases() if c[0] == "stream" and c[1] == "f16"}
    for c, n in names.items():
        assert n.startswith("dcn5_kernel<" if c[2] & 0x4000 else "dcn3_kernel<f16_t"), (c, n)
        assert n.endswith(", true>") == bool(c[2] & 0x18000 and not c[2] & 0x400), (c, n)
    if has_extra():
        assert {"dcn5_kernel<2, 2, 2, 256>", "dcn5_kernel<1, 2, 1, 256>", "dcn5_kernel<4, 4, 2, 256>"} <= set(names.values()), names
    else:       # the default library answers the 0x4000 request with a clear error instead of another kernel
        c = [c for c in DCN_CASES if c[0] == "stream" and c[1] == "f16" and c[2] & 0x4000][0]
        with pytest.raises(RuntimeError, match="EXTRA=1"):
            _dcn_built(c)[5].name


def test_dcn_auto_selection_reaches_mt4():
    c = [c for c in DCN_CASES if c[3] == 12 and c[0] == "fused" and c[1] == "bf16"][0]
    assert _dcn_built(c)[5].name == "dcn3_kernel<unsigned short, 4, 16, 2, 2, false, 0>"
    c = [c for c in DCN_CASES if c[3] == 12 and c[0] == "stream" and c[1] == "bf16"][0]
    assert _dcn_built(c)[5].name == "dcn3_kernel<unsigned short, 4, 16, 4, 2, true, 256>"
    c = [c for c in DCN_CASES if c[0] == "stream" and c[1] == "bf16" and c[2] == 0 and c[4:8] == (256, 256, 16, 16)][0]      # 1 tile x 2 groups: small grid
    assert _dcn_built(c)[5].name == "dcn3_kernel<unsigned short, 2, 16, 2, 2, true, 256>"


# ---- the bench plan's kernel set -----------------------------------------------------------------------------
def _plan_kernel_names(batch):
    import h3d_amd  # noqa: F401
    from h3d_amd import arch, synth
    from h3d_amd.detector import MultiPoseDetector, Opt
    opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    plan = det.model.engine(torch.device(DEV)).plan(batch, 512, 512)
    return {kernel_name(op) for op in plan.ops}


def test_bench_plan_kernels_are_all_covered():
    """tested-kernel-set >= bench-kernel-set: every conv / DeformConv instantiation of the batch-64 (bench.py,
    BASELINE configs[2]), batch-32 (configs[1]) and batch-8 (the 8-GPU shard of configs[2]) plans has a per-op parity case
    above or in test_gpu_conv.py;
    the remaining kernels (stem3, heads, max-pool, up-sample) have exactly one instantiation per dtype and are
    compared at full size in test_gpu_fullsize.py."""
    import test_gpu_conv
    from gpu_helpers import conv as _  # noqa: F401
    tested = {_conv2_built(c)[4].name for c in CONV2_CASES} | {_dcn_built(c)[5].name for c in _active_dcn_cases()}
    tested |= test_gpu_conv.conv_case_kernel_names("bf16") | test_gpu_conv.gemm1_case_kernel_names()
    single = ("stem3_kernel", "heads_kernel<", "maxpool_kernel<", "upadd_kernel<", "copy_kernel<")
    for batch in (64, 32, 16, 8):   # 16 / 8: the shards one GPU of four / eight gets from the headline batch (bench.py --global-batch 64)
        names = _plan_kernel_names(batch)
        missing = sorted(n for n in names if n not in tested and not n.startswith(single))
        assert not missing, "batch %d: no per-op parity case dispatches to %s" % (batch, missing)
        torch.cuda.empty_cache()
