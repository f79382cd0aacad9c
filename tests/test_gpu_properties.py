"""Size-independent properties at BASELINE.json's full sizes, through the product path (no oracle: it cannot run these sizes in
seconds).  They hold for any correct implementation of the reference functions and break for the classic full-grid bugs -- a
tile mapping that drops or doubles a tile, an index that overflows 32 bits, a race between workgroups:

  DeformConv (dcn_v2_cuda.cu:43-173)   linear in the input for fixed offsets / masks; zero mask -> bias only
  _nms / _topk (decode.py:6-41)        idempotent; scores sorted, equal to heat[index]; indices distinct; a batch permutation
                                       permutes the results
  network + decode                     the same image at every batch position of a full batch gives the same bits
"""
import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import arch, decode, synth
from h3d_amd.dcn_v2 import dcn_v2_forward
from h3d_amd.detector import MultiPoseDetector, Opt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dcn_operands(B, C, Co, H, W, seed, offset_scale):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * (1.0 / np.sqrt(9 * C))
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * offset_scale
    m = torch.sigmoid(torch.randn(B, 9, H, W, generator=g))
    return [t.to(DEV) for t in (x, w, b, off, m)]


def _dcn(x, w, b, off, m):
    return dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize("B,C,Co,H,W", [(8, 64, 64, 128, 128), (8, 128, 64, 64, 64), (4, 256, 256, 32, 32)])
def test_deform_conv_operator_is_linear_in_the_input_at_full_layer_size(B, C, Co, H, W):
    # the DLAUp / IDAUp DeformConv shapes of the 512 x 512 plan, fp32 operator boundary, mean |offset| 2.4 px
    x1, w, b, off, m = _dcn_operands(B, C, Co, H, W, 11, 3.0)
    x2 = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(12)).to(DEV)
    y1, y2 = _dcn(x1, w, b, off, m), _dcn(x2, w, b, off, m)
    bias = b.view(1, -1, 1, 1)
    y12 = _dcn(x1 + x2, w, b, off, m)
    scale = float(y12.abs().max())
    assert float((y12 - (y1 + y2 - bias)).abs().max()) < 2e-5 * max(scale, 1.0)      # fp32 sums in another order
    ya = _dcn(-2.5 * x1, w, b, off, m)
    assert float((ya - (-2.5 * (y1 - bias) + bias)).abs().max()) < 2e-5 * max(scale, 1.0)
    # zero mask: every sample is weighted 0 -> the bias alone, exactly
    y0 = _dcn(x1, w, b, off, torch.zeros_like(m))
    assert torch.equal(y0, bias.expand_as(y0).contiguous())
    # offsets that leave the image entirely: zero contribution as well (dcn_v2_im2col_cuda.cu:150 gate)
    yo = _dcn(x1, w, b, torch.full_like(off, 1.0e4), m)
    assert torch.equal(yo, bias.expand_as(yo).contiguous())


def test_nms_topk_properties_at_bench_size():
    # 64 images x (1 + 17) maps of 128 x 128, K = 100: the decode stage of BASELINE configs[2]
    B, K = 64, 100
    h = synth.synth_heads(B, 128, 128, 17, seed=5)
    for name in ("hm", "hm_hp"):
        heat = torch.from_numpy(h[name]).to(DEV)
        nms = decode._nms(heat)
        assert torch.equal(decode._nms(nms), nms)                            # idempotent
        assert bool(((nms == heat) | (nms == 0)).all())                      # keeps a value or zeroes it
        s, i, y, x = decode._topk_channel(nms, K)
        flat = nms.view(B, nms.shape[1], -1)
        assert torch.equal(torch.gather(flat, 2, i), s)                      # a score IS the map's value at its index
        assert bool((s[..., :-1] >= s[..., 1:]).all())                       # sorted, descending
        srt = torch.sort(i, dim=2).values
        assert bool((srt[..., 1:] != srt[..., :-1]).all())                   # K distinct pixels per map
        assert torch.equal(y, (i // 128).float()) and torch.equal(x, (i % 128).float())
        kth = s[..., -1:]
        assert bool(((flat > kth).sum(dim=2) <= K - 1).all())                # nothing larger than the K-th score was left out
        # ties: among equal scores the lowest index first (the oracle's and the golden vectors' rule)
        same = s[..., 1:] == s[..., :-1]
        assert bool((~same | (i[..., 1:] > i[..., :-1])).all())
        # batch permutation equivariance
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
        s2, i2, _, _ = decode._topk_channel(nms[perm].contiguous(), K)
        assert torch.equal(s2, s[perm]) and torch.equal(i2, i[perm])


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_batch_of_one_image_is_identical_at_every_batch_position(dtype):
    # 64 copies of one 512 x 512 image through the throughput plan + decode: 4096 DeformConv tiles, 2048 head tiles, 1152 heat maps --
    # every batch position must produce the bits of position 0 (the per-image work is independent: trainer.py:176)
    opt = Opt(input_h=512, input_w=512, smpl=True, dtype=dtype, smpl_people=4)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=torch.device(DEV))
    img = torch.from_numpy(synth.synth_image_batch(1, 512, 512, seed=317)).to(DEV)
    res = det.run(img.expand(64, 3, 512, 512).contiguous())
    for k in ("dets", "inds", "verts"):
        v = res[k]
        assert torch.equal(v, v[:1].expand_as(v)), k
    for k, v in res["heads"].items():
        assert torch.equal(v, v[:1].expand_as(v)), k
    assert bool(torch.isfinite(res["dets"]).all()) and bool(torch.isfinite(res["verts"]).all())
