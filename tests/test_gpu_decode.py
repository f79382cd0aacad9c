"""GPU parity of the decode kernels: bit-exact against (a) the golden vectors = the reference's
own outputs (tests/golden/decode_*.npz) on the tie-free prefix and (b) the oracle restatement
everywhere (same tie rule: lowest index first)."""
import os

import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import decode, synth, utils
from h3d_amd.detector import multi_pose_post_process
from oracle import decode as odec
from oracle import post_process as opost
from test_oracle_golden import CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(h):
    return {k: torch.from_numpy(v).to(DEV) for k, v in h.items()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_multi_pose_decode_bit_exact(golden_dir, name):
    B, H, W, K, seed, use_reg, use_hp, use_off = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    h = synth.synth_heads(B, H, W, 17, seed)
    d = _dev(h)
    dets = decode.multi_pose_decode(d["hm"], d["wh"], d["hps"], reg=d["reg"] if use_reg else None,
                                    hm_hp=d["hm_hp"] if use_hp else None,
                                    hp_offset=d["hp_offset"] if use_off else None, K=K).cpu().numpy()
    # (b) oracle: identical everywhere, ties included
    ref = odec.multi_pose_decode(h["hm"], h["wh"], h["hps"], reg=h["reg"] if use_reg else None,
                                 hm_hp=h["hm_hp"] if use_hp else None,
                                 hp_offset=h["hp_offset"] if use_off else None, K=K)
    np.testing.assert_array_equal(dets, ref)
    # (a) the reference's own outputs on the tie-free prefix
    for b in range(B):
        p = odec.strict_prefix(g["topk_scores"][b], K)
        np.testing.assert_array_equal(dets[b, :p], g["dets"][b, :p])
        np.testing.assert_array_equal(dets[b, :, 4], g["dets"][b, :, 4])      # scores: all rows


@pytest.mark.parametrize("name", ["decode_128x128_k100", "decode_16x24_k100_tied", "decode_48x64_k100"])
def test_topk_indices_bit_exact(golden_dir, name):
    B, H, W, K, seed = CASES[name][:5]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    h = synth.synth_heads(B, H, W, 17, seed)
    heat = decode._nms(torch.from_numpy(h["hm"]).to(DEV))
    np.testing.assert_array_equal(heat.cpu().numpy(), odec.nms(h["hm"]))
    s, i, c, y, x = [t.cpu().numpy() for t in decode._topk(heat, K)]
    rs, ri, rc, ry, rx = odec.topk(odec.nms(h["hm"]), K)
    np.testing.assert_array_equal(s, rs)
    np.testing.assert_array_equal(i, ri)
    np.testing.assert_array_equal(c, rc)
    np.testing.assert_array_equal(y, ry)
    np.testing.assert_array_equal(x, rx)
    assert i.dtype == np.int64 and c.dtype == np.int32
    np.testing.assert_array_equal(s, g["topk_scores"])
    for b in range(B):
        p = odec.strict_prefix(g["topk_scores"][b], K)
        np.testing.assert_array_equal(i[b, :p], g["topk_inds"][b, :p])
    hp = decode._nms(torch.from_numpy(h["hm_hp"]).to(DEV))
    hs, hi, hy, hx = [t.cpu().numpy() for t in decode._topk_channel(hp, K)]
    os_, oi, oy, ox = odec.topk_channel(odec.nms(h["hm_hp"]), K)
    np.testing.assert_array_equal(hs, os_)
    np.testing.assert_array_equal(hi, oi)
    np.testing.assert_array_equal(hs, g["hp_scores"])
    for b in range(B):
        for j in range(17):
            q = odec.strict_prefix(g["hp_scores"][b, j], K)
            np.testing.assert_array_equal(hi[b, j, :q], g["hp_inds"][b, j, :q])


def test_ctdet_decode_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "ctdet_32x32_c80.npz"))
    B, C, H, W, K = 2, 80, 32, 32, 100
    u = synth.uniform("ctdet_hm", (B, C, H, W), 0.0, 1.0, 7)
    hm = np.clip((u * u) * (u * u) * np.float32(0.9), np.float32(1e-4), np.float32(1 - 1e-4)).astype(np.float32)
    wh = synth.uniform("ctdet_wh", (B, 2, H, W), 2.0, 20.0, 7)
    reg = synth.uniform("ctdet_reg", (B, 2, H, W), 0.0, 1.0, 7)
    dets = decode.ctdet_decode(torch.from_numpy(hm).to(DEV), torch.from_numpy(wh).to(DEV),
                               reg=torch.from_numpy(reg).to(DEV), K=K).cpu().numpy()
    np.testing.assert_array_equal(dets, g["dets"])


def test_all_equal_map_ties_lowest_index_first():
    hm = torch.full((1, 1, 16, 16), 0.5, device=DEV)
    s, i, c, y, x = decode._topk(decode._nms(hm), 10)
    assert i.cpu().tolist() == [list(range(10))]
    assert s.cpu().tolist() == [[0.5] * 10]


def test_paired_topk_launch_matches_two_launches():
    # h3d_nms_topk2 (hm + hm_hp in one launch: decode._map_topk2) vs h3d_nms_topk per tensor: the same workgroup
    # program on the same maps -> identical scores, indices and coordinates (ragged map size, both flag settings)
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(3, 1, 24, 40, generator=g) * 2 - 1).to(DEV)
    b = (torch.randn(3, 17, 24, 40, generator=g) * 2 - 1).to(DEV)
    for flags in (0, decode.NMS_SIGMOID):
        pa, pb = decode._map_topk2(a, b, 20, flags)
        for got, ref in ((pa, decode._map_topk(a, 20, flags)), (pb, decode._map_topk(b, 20, flags))):
            for x, y in zip(got, ref):
                assert torch.equal(x, y)


@pytest.mark.parametrize("shape", [(2, 3, 184, 320), (1, 17, 200, 200), (1, 1, 400, 352)])
def test_maps_larger_than_the_lds_go_through_bands_and_match_the_oracle(shape):
    """H * W > 36864 (VERDICT r2 'missing' 4: the 320 x 184 output of a --keep_res 1280 x 736 frame, datasets/coco.py:160-163):
    h3d_nms_topk_large cuts a map into bands of rows (+ halo rows for the 3x3 max) and merges their top K.  Bit-exact against the
    oracle's _nms + _topk_channel (numpy, same tie rule) -- peaks ON band boundaries, plateaus of equal scores across bands and
    a map with fewer than K positive peaks included -- and the full multi_pose decode at that size against the oracle's."""
    from oracle import decode as odec
    B, C, H, W = shape
    K = 100
    rng = np.random.default_rng(7)
    heat = rng.random((B, C, H, W), dtype=np.float32) ** 8
    heat[:, :, :, :8] = 0.25                               # a plateau crossing every band: ties resolved by flat index
    heat[0, 0] = 0.0
    heat[0, 0, 5::37, 3::41] = 0.9                          # < K isolated peaks, the rest zeros: zeros fill in index order
    heat = np.clip(heat, 1e-4, 1 - 1e-4).astype(np.float32)
    s, i, y, x = decode._map_topk(torch.from_numpy(heat).to(DEV), K, 0)
    rs, ri, ry, rx = odec.topk_channel(odec.nms(heat), K)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    np.testing.assert_array_equal(y.cpu().numpy(), ry)
    np.testing.assert_array_equal(x.cpu().numpy(), rx)
    if C == 17:
        h = synth.synth_heads(B, H, W, 17, 3)
        d = _dev(h)
        dets = decode.multi_pose_decode(d["hm"], d["wh"], d["hps"], d["reg"], d["hm_hp"], d["hp_offset"], K=K).cpu().numpy()
        ref = odec.multi_pose_decode(h["hm"], h["wh"], h["hps"], h["reg"], h["hm_hp"], h["hp_offset"], K=K)
        np.testing.assert_array_equal(dets, ref)


def test_topk_k_out_of_range_raises():
    hm = torch.zeros(1, 1, 4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="out of range"):
        decode._topk(hm, 17)


def test_sigmoid_and_gather(golden_dir):
    g = np.load(os.path.join(golden_dir, "sigmoid.npz"))
    y = utils._sigmoid(torch.from_numpy(g["x"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1.2e-7)
    assert y.min() == np.float32(1e-4) and y.max() == np.float32(1 - 1e-4)
    feat = torch.from_numpy(synth.uniform("f", (2, 5, 6, 7), -1, 1)).to(DEV)
    ind = torch.tensor([[0, 41, 7], [3, 3, 20]], device=DEV)
    got = utils._transpose_and_gather_feat(feat, ind).cpu()
    ref = feat.cpu().permute(0, 2, 3, 1).reshape(2, 42, 5).gather(1, ind.cpu().unsqueeze(2).expand(2, 3, 5))
    assert torch.equal(got, ref)
    got2 = utils._gather_feat(feat.permute(0, 2, 3, 1).reshape(2, 42, 5).contiguous(), ind).cpu()
    assert torch.equal(got2, ref)


def test_fused_logits_decode_matches_two_step():
    # detector path (sigmoid folded into the NMS kernel) == _sigmoid then multi_pose_decode
    h = synth.synth_heads(2, 32, 48, 17, 9)
    d = _dev(h)
    logit = lambda p: torch.log(p / (1 - p))
    hm_l, hp_l = logit(d["hm"]), logit(d["hm_hp"])
    a = decode.multi_pose_decode_logits(hm_l, d["wh"], d["hps"], d["reg"], hp_l, d["hp_offset"], K=30)
    b = decode.multi_pose_decode(utils._sigmoid(hm_l), d["wh"], d["hps"], d["reg"], utils._sigmoid(hp_l),
                                 d["hp_offset"], K=30)
    assert torch.equal(a, b)


def test_post_process_vs_oracle():
    h = synth.synth_heads(2, 128, 128, 17, 0)
    d = _dev(h)
    dets = decode.multi_pose_decode(d["hm"], d["wh"], d["hps"], d["reg"], d["hm_hp"], d["hp_offset"], K=100)
    c = np.array([[320.0, 240.0], [500.5, 333.25]], np.float32)
    s = np.array([640.0, 1001.0], np.float32)
    got = multi_pose_post_process(dets, c, s, 128, 128).cpu().numpy()
    ref = np.stack(opost.multi_pose_post_process(dets.cpu().numpy(), c, s, 128, 128))
    ok = np.abs(ref) < 1e5                                   # -10000-sentinel rows stay huge; compare relatively
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5, atol=2e-3)


def test_full_size_batch_properties():
    # BASELINE size (B=64, 128x128 maps, K=100): properties that need no oracle at this size
    B = 64
    h = synth.synth_heads(B, 128, 128, 17, 11)
    d = _dev(h)
    dets, aux = decode._multi_pose(d["hm"], d["wh"], d["hps"], d["reg"], d["hm_hp"], d["hp_offset"], 100, False, True)
    s = aux["scores"].cpu().numpy()
    assert (np.diff(s, axis=1) <= 0).all()                               # sorted descending
    inds = aux["inds"].cpu().numpy()
    assert all(len(set(r)) == 100 for r in inds)                         # distinct peaks
    flat = torch.from_numpy(h["hm"]).reshape(B, -1)
    np.testing.assert_array_equal(np.take_along_axis(flat.numpy(), inds, 1), s)   # score == heat at index
    # batch-position invariance: image 5 alone decodes to the same detections
    one = decode.multi_pose_decode(d["hm"][5:6], d["wh"][5:6], d["hps"][5:6], d["reg"][5:6], d["hm_hp"][5:6],
                                   d["hp_offset"][5:6], K=100)
    assert torch.equal(one[0], dets[5])
    # oracle spot check on 2 of the 64 images
    ref = odec.multi_pose_decode(h["hm"][:2], h["wh"][:2], h["hps"][:2], h["reg"][:2], h["hm_hp"][:2],
                                 h["hp_offset"][:2], K=100)
    np.testing.assert_array_equal(dets[:2].cpu().numpy(), ref)


def test_ctdet_task_entry_end_to_end():
    # the `ctdet` branch of the task dispatch (trains/trainer.py:444-455): heads of opts.py:241-247 through the same
    # engine, `_sigmoid` + ctdet_decode bit-exact against the oracle on the GPU's own heads, ctdet_post_process
    # (utils/post_process.py:24-38) against its restatement
    from h3d_amd import arch
    from h3d_amd.detector import CtdetDetector, MultiPoseDetector, Opt, ctdet_post_process, make_detector
    opt = Opt(task="ctdet", input_h=128, input_w=160, dtype="f32", K=40, num_classes=80)
    assert opt.heads == {"hm": 80, "wh": 2, "reg": 2}
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
    det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    assert isinstance(det, CtdetDetector)
    assert isinstance(make_detector(Opt(task="multi_pose"), device=DEV), MultiPoseDetector)
    with pytest.raises(ValueError, match="task not defined"):
        Opt(task="ddd")
    xs = torch.from_numpy(synth.synth_images(2, 128, 160, seed=5)).to(DEV)
    c = np.array([[80.0, 64.0], [100.5, 70.25]], np.float32)
    s = np.array([160.0, 300.0], np.float32)
    res = det.run(xs, meta={"c": c, "s": s})
    heads = {k: v.cpu().numpy() for k, v in res["heads"].items()}
    # (the device's expf and numpy's differ by an ulp on some scores: decode the GPU's own _sigmoid output, whose parity
    #  with the reference's is pinned by the golden sigmoid.npz in test_sigmoid_and_gather)
    ref = odec.ctdet_decode(utils._sigmoid(res["heads"]["hm"].clone()).cpu().numpy(), heads["wh"], heads["reg"], K=40)
    np.testing.assert_array_equal(res["dets"].cpu().numpy(), ref)
    exp = opost.ctdet_post_process(ref, c, s, 32, 40, 80)
    got = res["results"]
    assert len(got) == 2 and set(got[0]) == set(range(1, 81))
    for i in range(2):
        for k in range(1, 81):
            assert len(got[i][k]) == len(exp[i][k])
            if exp[i][k]:
                np.testing.assert_allclose(np.array(got[i][k]), np.array(exp[i][k]), rtol=1e-5, atol=2e-3)
    assert sum(len(v) for v in got[0].values()) == 40
    again = ctdet_post_process(res["dets"], c, s, 32, 40, 80)
    assert again == got


def test_gather_and_flip_helpers_match_reference_golden(golden_dir):
    # h3d_gather_feat / the device-resident flip helpers against OUTPUTS OF THE REFERENCE's models/utils.py:12-51
    from test_oracle_golden import FLIP_IDX, _utils_inputs
    g = np.load(os.path.join(golden_dir, "utils_flip_gather.npz"))
    hm, hps, feat, ind = _utils_inputs()
    fd, idd = torch.from_numpy(feat).to(DEV), torch.from_numpy(ind).to(DEV)
    assert np.array_equal(utils._transpose_and_gather_feat(fd, idd).cpu().numpy(), g["transpose_and_gather"])
    assert np.array_equal(utils._gather_feat(fd.permute(0, 2, 3, 1).reshape(2, 42, 5).contiguous(), idd).cpu().numpy(), g["gather"])
    assert np.array_equal(utils.flip_tensor(torch.from_numpy(hm).to(DEV)).cpu().numpy(), g["flip_tensor"])
    assert np.array_equal(utils.flip_lr(torch.from_numpy(hm).to(DEV), FLIP_IDX).cpu().numpy(), g["flip_lr"])
    assert np.array_equal(utils.flip_lr_off(torch.from_numpy(hps).to(DEV), FLIP_IDX).cpu().numpy(), g["flip_lr_off"])


def test_single_class_topk_shortcut_equals_the_merge_kernel(golden_dir):
    # multi_pose has ONE class (opts.py:248): stage 2 of `_topk` over one class is the identity, so `_merge` returns views instead of
    # launching topk_merge_kernel -- the same bits as the kernel, also on a map full of ties, and the same dets end to end
    from h3d_amd import decode as dec
    for name, (B, H, W) in (("decode_16x24_k100_tied", (2, 16, 24)), ("decode_128x128_k100", (2, 128, 128))):
        h = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synth_heads(B, H, W, 17, 2 if "tied" in name else 0).items()}
        outs = []
        for flag in (True, False):
            dec.SINGLE_CLASS_SHORTCUT = flag
            try:
                t = dec._topk(dec._nms(h["hm"]), K=100)
                d = dec.multi_pose_decode(h["hm"], h["wh"], h["hps"], reg=h["reg"], hm_hp=h["hm_hp"], hp_offset=h["hp_offset"], K=100)
            finally:
                dec.SINGLE_CLASS_SHORTCUT = True
            outs.append([x.clone() for x in t] + [d.clone()])
        for a, b in zip(*outs):
            assert a.dtype == b.dtype and torch.equal(a, b), name


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_ctdet_end_to_end_f32_vs_reference_output(golden_dir, dtype):
    """(f16x3, round 5: the same contract on the fp16 matrix cores -- fp32 storage, three fp16 MFMAs on split operands per fp32 product.)
    The `ctdet` task entry against reference OUTPUT (tests/golden/e2e_ctdet_256.npz: the imported reference's plain-conv
    `dla_net` with the ctdet heads -> `_sigmoid` -> `ctdet_decode(K=100)`, trainer.py:444-455): the product's `CtdetDetector`
    (f32 plan, `not_use_dcn=True`) returns the reference's top-100 indices and classes on both images and its `dets` within
    2e-3 px / 2e-6 of score."""
    import os
    from h3d_amd import arch
    from h3d_amd.detector import Opt, make_detector
    g = np.load(os.path.join(golden_dir, "e2e_ctdet_256.npz"))
    opt = Opt(task="ctdet", input_h=256, input_w=256, dtype=dtype, K=100, num_classes=80, not_use_dcn=True)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, False), seed=0, gain=1.1)
    det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    res = det.run(torch.from_numpy(synth.synth_images(2, 256, 256, seed=317)).to(DEV))
    heads = {k: v.cpu().numpy() for k, v in res["heads"].items()}
    assert float(np.abs(heads["hm"][:, :, ::2, ::2] - g["hm_s2"]).max()) <= 2e-4
    for k in ("wh", "reg"):
        assert float(np.abs(heads[k] - g[k]).max()) <= 2e-4, k
    from h3d_amd import decode as dec
    s, inds, clses, ys, xs = dec._topk(dec._nms(utils._sigmoid(res["heads"]["hm"].clone())), K=100)
    np.testing.assert_array_equal(inds.cpu().numpy(), g["topk_inds"])
    np.testing.assert_array_equal(clses.cpu().numpy(), g["topk_clses"])
    np.testing.assert_allclose(s.cpu().numpy(), g["topk_scores"], rtol=0, atol=2e-6)
    d = res["dets"].cpu().numpy()
    np.testing.assert_allclose(d, g["dets"], rtol=0, atol=2e-3)
    np.testing.assert_array_equal(d[..., 5], g["dets"][..., 5])
