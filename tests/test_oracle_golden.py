"""CPU: the oracle restatements (oracle/dla.py, oracle/decode.py) against golden vectors that
are OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/*.npz, made by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import synth
from oracle import decode as odec
from oracle import dla as odla

HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}

CASES = {
    "decode_128x128_k100": (2, 128, 128, 100, 0, True, True, True),
    "decode_48x64_k100": (3, 48, 64, 100, 1, True, True, True),
    "decode_16x24_k100_tied": (2, 16, 24, 100, 2, True, True, True),
    "decode_64x64_k40": (1, 64, 64, 40, 3, True, True, True),
    "decode_32x32_noreg": (2, 32, 32, 50, 4, False, True, False),
    "decode_32x32_nohp": (1, 32, 32, 50, 5, True, False, False),
}


def test_sigmoid_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "sigmoid.npz"))
    y = odec.sigmoid_clamp(g["x"])
    # numpy exp vs torch's vectorised sigmoid may differ by an ulp; the clamp plateaus are exact
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1.2e-7)
    assert (y[g["y"] == np.float32(1e-4)] == np.float32(1e-4)).all()


def test_state_dict_shape_table_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "dla34_shapes.json")))
    ours = odla.state_dict_shapes(HEADS, use_dcn=False)
    assert set(ours) == set(ref)
    for k in ref:
        assert tuple(ours[k]) == tuple(ref[k]), k


def test_dla34_plain_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "dla34_plain.npz"))
    shapes = odla.state_dict_shapes(HEADS, use_dcn=False)
    sd = synth.synth_state_dict(shapes, seed=0)
    net = odla.DLAOracle(sd, HEADS, use_dcn=False)
    x = torch.from_numpy(synth.synth_images(2, 96, 128, seed=317))
    with torch.no_grad():
        out = net(x)[0]
    for k in HEADS:
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-5, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("name", sorted(CASES))
def test_multi_pose_decode_matches_reference(golden_dir, name):
    B, H, W, K, seed, use_reg, use_hp, use_off = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    h = synth.synth_heads(B, H, W, 17, seed)
    dets, aux = odec.multi_pose_decode(
        h["hm"], h["wh"], h["hps"], reg=h["reg"] if use_reg else None,
        hm_hp=h["hm_hp"] if use_hp else None, hp_offset=h["hp_offset"] if use_off else None,
        K=K, return_aux=True)
    # scores are identical as multisets everywhere; indices are bit-exact on the strict prefix
    np.testing.assert_array_equal(aux["scores"], g["topk_scores"])
    for b in range(B):
        p = odec.strict_prefix(g["topk_scores"][b], K)
        assert p >= min(K, int(g["nms_hm_nonzero"][b])) - 1 or p == K
        np.testing.assert_array_equal(aux["inds"][b, :p], g["topk_inds"][b, :p])
        np.testing.assert_array_equal(aux["ys"][b, :p], g["topk_ys"][b, :p])
        np.testing.assert_array_equal(aux["xs"][b, :p], g["topk_xs"][b, :p])
        np.testing.assert_array_equal(aux["clses"][b, :p], g["topk_clses"][b, :p])
        if use_hp:
            np.testing.assert_array_equal(aux["hm_score"][b], g["hp_scores"][b])
            hp_strict = True
            for j in range(17):
                q = odec.strict_prefix(g["hp_scores"][b, j], K)
                np.testing.assert_array_equal(aux["hm_inds"][b, j, :q], g["hp_inds"][b, j, :q])
                hp_strict &= (q == K)
        else:
            hp_strict = True
        # detections: rows in the strict prefix; keypoint columns additionally need every
        # joint's candidate list to be tie-free (ties only ever involve sub-threshold 1e-4 / 0
        # scores, which the 0.1 threshold masks out, so in practice all rows agree)
        np.testing.assert_array_equal(dets[b, :p, :5], g["dets"][b, :p, :5])
        np.testing.assert_array_equal(dets[b, :p, 39], g["dets"][b, :p, 39])
        np.testing.assert_array_equal(dets[b, :p, 5:39], g["dets"][b, :p, 5:39])


def test_strict_prefix_covers_all_real_peaks(golden_dir):
    g = np.load(os.path.join(golden_dir, "decode_128x128_k100.npz"))
    for b in range(2):
        assert odec.strict_prefix(g["topk_scores"][b], 100) == 100


def test_ctdet_decode_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ctdet_32x32_c80.npz"))
    B, C, H, W, K = 2, 80, 32, 32, 100
    u = synth.uniform("ctdet_hm", (B, C, H, W), 0.0, 1.0, 7)
    hm = np.clip((u * u) * (u * u) * np.float32(0.9), np.float32(1e-4), np.float32(1 - 1e-4)).astype(np.float32)
    wh = synth.uniform("ctdet_wh", (B, 2, H, W), 2.0, 20.0, 7)
    reg = synth.uniform("ctdet_reg", (B, 2, H, W), 0.0, 1.0, 7)
    s, inds, clses, ys, xs = odec.topk(odec.nms(hm), K)
    np.testing.assert_array_equal(s, g["topk_scores"])
    for b in range(B):
        p = odec.strict_prefix(g["topk_scores"][b], K)
        assert p == K
        np.testing.assert_array_equal(inds[b], g["topk_inds"][b])
        np.testing.assert_array_equal(clses[b], g["topk_clses"][b])
    dets = odec.ctdet_decode(hm, wh, reg=reg, K=K)
    np.testing.assert_array_equal(dets, g["dets"])


FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def _utils_inputs():
    hm = synth.uniform("flip_hm", (2, 17, 6, 10), -1.0, 1.0, 11)
    hps = synth.uniform("flip_hps", (2, 34, 6, 10), -1.0, 1.0, 11)
    feat = synth.uniform("gather_feat", (2, 5, 6, 7), -1.0, 1.0, 11)
    ind = np.array([[0, 41, 7, 7], [3, 3, 20, 40]], dtype=np.int64)
    return hm, hps, feat, ind


def test_flip_and_gather_restatements_match_reference_outputs(golden_dir):
    # models/utils.py:12-51 run in the build container (oracle/gen_golden.py --utils-only): pins oracle/decode.py's
    # flip_tensor / flip_lr / flip_lr_off / gather_nchw bit for bit (pure data movement)
    g = np.load(os.path.join(golden_dir, "utils_flip_gather.npz"))
    hm, hps, feat, ind = _utils_inputs()
    assert np.array_equal(odec.flip_tensor(hm), g["flip_tensor"])
    assert np.array_equal(odec.flip_lr(hm, FLIP_IDX), g["flip_lr"])
    assert np.array_equal(odec.flip_lr_off(hps, FLIP_IDX), g["flip_lr_off"])
    assert np.array_equal(odec.gather_nchw(feat, ind), g["transpose_and_gather"])
    assert np.array_equal(g["gather"], g["transpose_and_gather"])
    # and the product's host helpers (torch, device-agnostic data movement) against the same fixtures
    from h3d_amd import utils
    assert np.array_equal(utils.flip_tensor(torch.from_numpy(hm)).numpy(), g["flip_tensor"])
    assert np.array_equal(utils.flip_lr(torch.from_numpy(hm), FLIP_IDX).numpy(), g["flip_lr"])
    assert np.array_equal(utils.flip_lr_off(torch.from_numpy(hps), FLIP_IDX).numpy(), g["flip_lr_off"])


def test_dla34_plain_forward_gain125_matches_reference(golden_dir):
    # the same 96x128 case with signal-preserving weights (head maps with O(1)..O(10) variation instead of a near-constant)
    g = np.load(os.path.join(golden_dir, "dla34_plain_g125.npz"))
    shapes = odla.state_dict_shapes(HEADS, use_dcn=False)
    sd = synth.synth_state_dict(shapes, seed=0, gain=1.25)
    net = odla.DLAOracle(sd, HEADS, use_dcn=False)
    with torch.no_grad():
        out = net(torch.from_numpy(synth.synth_images(2, 96, 128, seed=317)))[0]
    for k in HEADS:
        scale = float(np.abs(g[k]).max())
        assert scale > 1.0, (k, scale)
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=0, atol=2e-5 * scale, err_msg=k)


def test_end_to_end_images_to_indices_matches_reference(golden_dir):
    """The metric's second clause against reference OUTPUT: `e2e_plain_512.npz` is what the reference's own
    `dla_net(not_use_dcn=True)` -> `_sigmoid` -> `multi_pose_decode` (trainer.py:93,127,456-469) returns on two 512x512 images.
    The oracle restatement (oracle/dla.py + oracle/decode.py), from the images, must give the same heads (fp32 rounding
    of a different but equivalent op order) and -- on ITS OWN heads -- the same peak indices wherever the reference's
    score gaps exceed that difference; on the REFERENCE's logits the decode must be bit-identical."""
    from oracle import index_match as oim
    g = np.load(os.path.join(golden_dir, "e2e_plain_512.npz"))
    K = 100
    sd = synth.synth_state_dict(odla.state_dict_shapes(HEADS, use_dcn=False), seed=0, gain=1.25)
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    with torch.no_grad():
        out = {k: v.numpy() for k, v in odla.DLAOracle(sd, HEADS, use_dcn=False)(torch.from_numpy(synth.synth_images(2, 512, 512, seed=317)))[0].items()}
    for k in ("hm", "hm_hp"):
        np.testing.assert_allclose(out[k], g[k], rtol=0, atol=2e-5 * float(np.abs(g[k]).max()), err_msg=k)
    for k in ("wh", "hps", "reg", "hp_offset"):
        np.testing.assert_allclose(out[k][:, :, ::4, ::4], g[k + "_s4"], rtol=0, atol=2e-5 * float(np.abs(g[k + "_s4"]).max()), err_msg=k)
    # (a) decode restatement on the reference's own logits: scores as the reference's _sigmoid gives them (<= 1 ulp), indices
    #     bit-identical on the tie-free prefix (every real peak: ties only occur on the 1e-4 clamp plateau)
    hm_sig = odec.sigmoid_clamp(g["hm"])
    np.testing.assert_allclose(hm_sig, g["hm_sig"], rtol=0, atol=1.2e-7)
    s, inds, clses, ys, xs = odec.topk(odec.nms(g["hm_sig"]), K)
    np.testing.assert_array_equal(s, g["topk_scores"])
    for b in range(2):
        p = odec.strict_prefix(g["topk_scores"][b], K)
        assert p == K
        np.testing.assert_array_equal(inds[b], g["topk_inds"][b])
    # (b) images -> indices through the oracle network
    m = oim.index_match(out, odec.topk(odec.nms(odec.sigmoid_clamp(out["hm"])), K)[1], {"hm": g["hm"]}, K=K)
    assert m["robust_prefix_logit"] > 0 and m["robust_prefix_logit_equal"] and m["robust_prefix_equal"] and m["agreement"] >= 0.99, m
    hp = odec.sigmoid_clamp(out["hm_hp"])
    dets, aux = odec.multi_pose_decode(odec.sigmoid_clamp(out["hm"]), out["wh"], out["hps"], out["reg"], hp, out["hp_offset"], K=K, return_aux=True)
    same = (aux["inds"] == g["topk_inds"])
    assert same.mean() >= 0.99, same.mean()
    rows = same                                                         # rows whose centre index agrees: the detection row must too
    np.testing.assert_allclose(dets[rows][:, :5], g["dets"][rows][:, :5], rtol=0, atol=2e-3)
    # hm_hp: several joint maps of this random-weight network saturate at the 1 - 1e-4 clamp (whole plateaus of equal scores,
    # whose order torch leaves unspecified): scores agree everywhere, indices on each channel's tie-free prefix
    np.testing.assert_allclose(aux["hm_score"], g["hp_scores"], rtol=0, atol=2e-3)
    checked = 0
    for b in range(2):
        for j in range(17):
            q = odec.strict_prefix(g["hp_scores"][b, j], K)
            eq = aux["hm_inds"][b, j, :q] == g["hp_inds"][b, j, :q]
            checked += q
            assert q == 0 or eq.mean() >= 0.97, (b, j, q, eq.mean())
    assert checked >= 500, checked


def test_ctdet_end_to_end_matches_reference(golden_dir):
    """The `ctdet` task (trainer.py:444-455) against reference OUTPUT: e2e_ctdet_256.npz = the imported reference's
    `dla_net({'hm': 80, 'wh': 2, 'reg': 2}, not_use_dcn=True)` -> `_sigmoid` -> `ctdet_decode(K=100)` on two 256x256 images.
    The oracle network reproduces the heads; its decode ON THE REFERENCE'S HEADS reproduces `_topk` (indices, classes) and `dets`
    bit for bit (all 200 top scores are distinct)."""
    g = np.load(os.path.join(golden_dir, "e2e_ctdet_256.npz"))
    heads = {"hm": 80, "wh": 2, "reg": 2}
    sd = synth.synth_state_dict(odla.state_dict_shapes(heads, use_dcn=False), seed=0, gain=1.1)
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    with torch.no_grad():
        out = {k: v.numpy() for k, v in odla.DLAOracle(sd, heads, use_dcn=False)(torch.from_numpy(synth.synth_images(2, 256, 256, seed=317)))[0].items()}
    np.testing.assert_allclose(out["hm"][:, :, ::2, ::2], g["hm_s2"], rtol=0, atol=2e-5 * float(np.abs(g["hm_s2"]).max()))
    for k in ("wh", "reg"):
        np.testing.assert_allclose(out[k], g[k], rtol=0, atol=2e-5 * float(np.abs(g[k]).max()), err_msg=k)
    assert len(np.unique(g["topk_scores"])) == 200
    # decode on the oracle's own heads: same peaks (the head difference is 1e-5 of a logit, the score gaps are larger)
    hm = odec.sigmoid_clamp(out["hm"])
    s, inds, clses, ys, xs = odec.topk(odec.nms(hm), 100)
    np.testing.assert_array_equal(inds, g["topk_inds"])
    np.testing.assert_array_equal(clses, g["topk_clses"])
    np.testing.assert_allclose(s, g["topk_scores"], rtol=0, atol=2e-6)
    dets = odec.ctdet_decode(hm, out["wh"], reg=out["reg"], K=100)
    np.testing.assert_allclose(dets, g["dets"], rtol=0, atol=2e-3)
    np.testing.assert_array_equal(dets[..., 5], g["dets"][..., 5])
