#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- golden-vector generator.  Runs ONLY in the build container:
imports the reference's own Python code from /root/reference (read-only) and records its
OUTPUTS on deterministic synthetic inputs (h3d_amd.synth) into tests/golden/*.npz.
No reference source text is copied; the fixtures hold numbers only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What is recorded
  dla34_plain.npz      reference `dla_net(heads, not_use_dcn=True)` (models/model.py:501-516) --
                       the only variant the reference can run on a CPU -- on synthetic weights,
                       input [2,3,96,128]; outputs = the six multi_pose heads.
  dla34_shapes.json    the reference model's state_dict {key: shape} table.
  dla34_plain_g125.npz same model and input with the signal-preserving gain-1.25 weights (head maps with O(1) variation).
  e2e_plain_512.npz    END TO END, images -> indices, by the reference's own code in the reference's call order
                       (trains/trainer.py:93,127 `_sigmoid` on hm / hm_hp, then trainer.py:456-469 `multi_pose_decode`):
                       `dla_net(heads, not_use_dcn=True)` on synth_images(2, 512, 512, seed=317) with gain-1.25 weights;
                       records the raw `hm` / `hm_hp` logits, every 4th pixel of the other heads, `dets`, `_topk` and
                       `_topk_channel` outputs, and the NMS survivor counts.
  e2e_ctdet_256.npz    the `ctdet` task the same way (trainer.py:444-455): 80-class `hm`, `wh`, `reg` heads on 2 x 256 x 256 ->
                       `_sigmoid` -> `ctdet_decode(K=100)`: `dets`, `_topk` outputs, `wh` / `reg`, every 2nd pixel of `hm`.
  decode_*.npz         reference `_nms/_topk/_topk_channel/multi_pose_decode/ctdet_decode`
                       (models/decode.py) outputs on synthetic post-sigmoid heads.
  sigmoid.npz          reference `_sigmoid` (models/utils.py:8-10) on a logit ramp.
  utils_flip_gather.npz  reference `_gather_feat`, `_transpose_and_gather_feat` (models/utils.py:12-27) and the flip-test helpers
                       `flip_tensor`, `flip_lr`, `flip_lr_off` (models/utils.py:29-51) on synthetic maps (inputs are
                       regenerated from h3d_amd.synth by the tests; the fixture holds outputs only).
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/lib"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import h3d_amd  # noqa: E402,F401
from h3d_amd import synth  # noqa: E402

from models import decode as ref_decode  # noqa: E402  (reference)
from models import model as ref_model  # noqa: E402    (reference)
from models import utils as ref_utils  # noqa: E402    (reference)

OUT = os.path.join(ROOT, "tests", "golden")
HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}

DECODE_CASES = {
    # name: (B, H, W, K, seed, use_reg, use_hm_hp, use_hp_offset)
    "decode_128x128_k100": (2, 128, 128, 100, 0, True, True, True),
    "decode_48x64_k100": (3, 48, 64, 100, 1, True, True, True),
    "decode_16x24_k100_tied": (2, 16, 24, 100, 2, True, True, True),
    "decode_64x64_k40": (1, 64, 64, 40, 3, True, True, True),
    "decode_32x32_noreg": (2, 32, 32, 50, 4, False, True, False),
    "decode_32x32_nohp": (1, 32, 32, 50, 5, True, False, False),
}


def gen_dla():
    torch.manual_seed(0)
    m = ref_model.dla_net(HEADS, not_use_dcn=True).eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "dla34_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=0, sort_keys=True)
    sd = synth.synth_state_dict(shapes, seed=0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = synth.synth_images(2, 96, 128, seed=317)
    with torch.no_grad():
        out = m(torch.from_numpy(x))[0]
    np.savez_compressed(os.path.join(OUT, "dla34_plain.npz"),
                        **{k: v.numpy() for k, v in out.items()})
    print("dla34_plain:", {k: tuple(v.shape) for k, v in out.items()},
          "hm range", float(out["hm"].min()), float(out["hm"].max()))


def gen_dla_gain():
    """The 96x128 heads again with gain-1.25 weights: at gain 1.0 the maps reaching the neck are nearly constant
    (h3d_amd/synth.py), so this is the fixture that exercises the backbone with live signal."""
    torch.manual_seed(0)
    m = ref_model.dla_net(HEADS, not_use_dcn=True).eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=1.25)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = synth.synth_images(2, 96, 128, seed=317)
    with torch.no_grad():
        out = m(torch.from_numpy(x))[0]
    np.savez_compressed(os.path.join(OUT, "dla34_plain_g125.npz"), **{k: v.numpy() for k, v in out.items()})
    print("dla34_plain_g125: hm range", float(out["hm"].min()), float(out["hm"].max()), "std", float(out["hm"].std()))


def gen_e2e():
    """images -> heads -> _sigmoid -> multi_pose_decode, all by the imported reference, at the benchmark's image size."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m = ref_model.dla_net(HEADS, not_use_dcn=True).eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=1.25)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = synth.synth_images(2, 512, 512, seed=317)
    K = 100
    with torch.no_grad():
        out = m(torch.from_numpy(x))[0]
        rec = {"hm": out["hm"].numpy().copy(), "hm_hp": out["hm_hp"].numpy().copy()}
        for k in ("wh", "hps", "reg", "hp_offset"):
            rec[k + "_s4"] = out[k][:, :, ::4, ::4].contiguous().numpy().copy()
        out["hm"] = ref_utils._sigmoid(out["hm"])                        # trainer.py:93 (in place, like the loss module)
        out["hm_hp"] = ref_utils._sigmoid(out["hm_hp"])                  # trainer.py:127
        rec["hm_sig"] = out["hm"].numpy().copy()
        heat = ref_decode._nms(out["hm"])
        s, inds, clses, ys, xs = ref_decode._topk(heat, K=K)
        rec.update(topk_scores=s.numpy(), topk_inds=inds.numpy(), topk_clses=clses.numpy(), topk_ys=ys.numpy(), topk_xs=xs.numpy())
        rec["nms_hm_nonzero"] = np.array([(heat[b] != 0).sum().item() for b in range(2)])
        hs, hi, hy, hx = ref_decode._topk_channel(ref_decode._nms(out["hm_hp"]), K=K)
        rec.update(hp_scores=hs.numpy(), hp_inds=hi.numpy())
        dets = ref_decode.multi_pose_decode(out["hm"], out["wh"], out["hps"], reg=out["reg"], hm_hp=out["hm_hp"],
                                            hp_offset=out["hp_offset"], K=K)            # trainer.py:456-460
        rec["dets"] = dets.numpy()
    np.savez_compressed(os.path.join(OUT, "e2e_plain_512.npz"), **rec)
    print("e2e_plain_512: hm logits range", float(rec["hm"].min()), float(rec["hm"].max()), "std", float(rec["hm"].std()),
          "survivors", rec["nms_hm_nonzero"], "top score", s[:, 0].numpy(), "100th", s[:, -1].numpy())


def gen_e2e_ctdet():
    """The `ctdet` task end to end by the imported reference (trains/trainer.py:444-455): `dla_net({'hm': 80, 'wh': 2, 'reg': 2},
    not_use_dcn=True)` on synth_images(2, 256, 256, seed=317), gain-1.1 weights (at 1.15 and above the 80-class map saturates the
    1 - 1e-4 clamp on more than 100 pixels: every top score equal) -> `_sigmoid(hm)` (trainer.py:93) ->
    `ctdet_decode(hm, wh, reg=reg, K=100)` (decode.py:44-75)."""
    heads = {"hm": 80, "wh": 2, "reg": 2}
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m = ref_model.dla_net(heads, not_use_dcn=True).eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=1.1)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = synth.synth_images(2, 256, 256, seed=317)
    with torch.no_grad():
        out = m(torch.from_numpy(x))[0]
        rec = {"hm_s2": out["hm"][:, :, ::2, ::2].contiguous().numpy().copy(), "wh": out["wh"].numpy().copy(), "reg": out["reg"].numpy().copy()}
        hm = ref_utils._sigmoid(out["hm"])
        heat = ref_decode._nms(hm)
        s, inds, clses, ys, xs = ref_decode._topk(heat, K=100)
        rec.update(topk_scores=s.numpy(), topk_inds=inds.numpy(), topk_clses=clses.numpy())
        dets = ref_decode.ctdet_decode(hm, out["wh"], reg=out["reg"], K=100)
        rec["dets"] = dets.numpy()
    np.savez_compressed(os.path.join(OUT, "e2e_ctdet_256.npz"), **rec)
    print("e2e_ctdet_256: top score", s[:, 0].numpy(), "100th", s[:, -1].numpy(), "classes used", len(set(clses.numpy().ravel().tolist())))


def gen_decode():
    for name, (B, H, W, K, seed, use_reg, use_hm_hp, use_off) in DECODE_CASES.items():
        h = {k: torch.from_numpy(v) for k, v in synth.synth_heads(B, H, W, 17, seed).items()}
        rec = {}
        heat = ref_decode._nms(h["hm"])
        s, inds, clses, ys, xs = ref_decode._topk(heat, K=K)
        rec.update(topk_scores=s.numpy(), topk_inds=inds.numpy(), topk_clses=clses.numpy(),
                   topk_ys=ys.numpy(), topk_xs=xs.numpy())
        rec["nms_hm_nonzero"] = np.array([(heat[b] != 0).sum().item() for b in range(B)])
        if use_hm_hp:
            hs, hi, hy, hx = ref_decode._topk_channel(ref_decode._nms(h["hm_hp"]), K=K)
            rec.update(hp_scores=hs.numpy(), hp_inds=hi.numpy(), hp_ys=hy.numpy(), hp_xs=hx.numpy())
        dets = ref_decode.multi_pose_decode(
            h["hm"].clone(), h["wh"].clone(), h["hps"].clone(),
            reg=h["reg"].clone() if use_reg else None,
            hm_hp=h["hm_hp"].clone() if use_hm_hp else None,
            hp_offset=h["hp_offset"].clone() if use_off else None, K=K)
        rec["dets"] = dets.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "dets", tuple(dets.shape), "survivors", rec["nms_hm_nonzero"])

    # ctdet (two-stage top-k with 80 classes)
    B, C, H, W, K = 2, 80, 32, 32, 100
    u = synth.uniform("ctdet_hm", (B, C, H, W), 0.0, 1.0, 7)
    hm = np.clip((u * u) * (u * u) * np.float32(0.9), np.float32(1e-4), np.float32(1 - 1e-4)).astype(np.float32)
    wh = synth.uniform("ctdet_wh", (B, 2, H, W), 2.0, 20.0, 7)
    reg = synth.uniform("ctdet_reg", (B, 2, H, W), 0.0, 1.0, 7)
    heat = ref_decode._nms(torch.from_numpy(hm))
    s, inds, clses, ys, xs = ref_decode._topk(heat, K=K)
    dets = ref_decode.ctdet_decode(torch.from_numpy(hm), torch.from_numpy(wh),
                                   reg=torch.from_numpy(reg), K=K)
    np.savez_compressed(os.path.join(OUT, "ctdet_32x32_c80.npz"), dets=dets.numpy(),
                        topk_scores=s.numpy(), topk_inds=inds.numpy(), topk_clses=clses.numpy(),
                        topk_ys=ys.numpy(), topk_xs=xs.numpy())
    print("ctdet", tuple(dets.shape))


def gen_sigmoid():
    x = np.concatenate([np.linspace(-20, 20, 4001, dtype=np.float32),
                        synth.normalish("sig", (4096,), -2.19, 3.0, 0)])
    y = ref_utils._sigmoid(torch.from_numpy(x.copy()))
    np.savez_compressed(os.path.join(OUT, "sigmoid.npz"), x=x, y=y.numpy())


FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]      # opts.py:250 (COCO left/right joints)


def gen_utils():
    hm = torch.from_numpy(synth.uniform("flip_hm", (2, 17, 6, 10), -1.0, 1.0, 11))
    hps = torch.from_numpy(synth.uniform("flip_hps", (2, 34, 6, 10), -1.0, 1.0, 11))
    feat = torch.from_numpy(synth.uniform("gather_feat", (2, 5, 6, 7), -1.0, 1.0, 11))
    ind = torch.tensor([[0, 41, 7, 7], [3, 3, 20, 40]], dtype=torch.int64)
    rec = {
        "flip_tensor": ref_utils.flip_tensor(hm).numpy(),
        "flip_lr": ref_utils.flip_lr(hm, FLIP_IDX).numpy(),
        "flip_lr_off": ref_utils.flip_lr_off(hps, FLIP_IDX).numpy(),
        "transpose_and_gather": ref_utils._transpose_and_gather_feat(feat, ind).numpy(),
        "gather": ref_utils._gather_feat(feat.permute(0, 2, 3, 1).contiguous().view(2, 42, 5), ind).numpy(),
    }
    np.savez_compressed(os.path.join(OUT, "utils_flip_gather.npz"), **rec)
    print("utils_flip_gather:", {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--utils-only" in sys.argv:
        gen_utils()
        raise SystemExit(0)
    if "--e2e-only" in sys.argv:
        gen_dla_gain()
        gen_e2e()
        gen_e2e_ctdet()
        raise SystemExit(0)
    if "--ctdet-only" in sys.argv:
        gen_e2e_ctdet()
        raise SystemExit(0)
    gen_sigmoid()
    gen_utils()
    gen_decode()
    gen_dla()
    gen_dla_gain()
    gen_e2e()
    gen_e2e_ctdet()
    print("golden vectors written to", OUT)
