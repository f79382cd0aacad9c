"""GPU: h3d_preprocess (csrc/preprocess.hip) bit-exact against the numpy restatement of the reference's
val pre-process (oracle/preprocess.py), through the C ABI."""
import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import preprocess as pre
from oracle import preprocess as opre

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("h,w,res", [(512, 512, 512), (480, 640, 512), (427, 640, 512), (333, 250, 256), (37, 91, 128)])
def test_preprocess_bit_exact(h, w, res):
    rng = np.random.default_rng(h * 1000 + w)
    imgs = rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
    out, c, s = pre.pre_process(torch.from_numpy(imgs).to(DEV), input_res=res)
    out = out.cpu().numpy()
    for b in range(3):
        want, wc, ws = opre.get_input(imgs[b], res=res)
        assert np.array_equal(out[b], want), (b, float(np.abs(out[b] - want).max()))
        assert np.array_equal(c[b], wc) and s[b] == ws


def test_preprocess_rejects_bad_input():
    with pytest.raises(ValueError):
        pre.pre_process(torch.zeros(1, 8, 8, 3, device=DEV))            # not uint8


def test_frames_to_image_pixel_results():
    # raw frames -> pre-process -> network -> decode -> post-process (results in original-image pixels)
    from h3d_amd import arch, synth
    from h3d_amd.detector import MultiPoseDetector, Opt, run_frames
    opt = Opt(input_h=128, input_w=128, smpl=False, dtype="bf16", K=20)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    frames = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(2, 96, 160, 3), dtype=np.uint8)).to(DEV)
    res = run_frames(det, frames)
    assert tuple(res["dets"].shape) == (2, 20, 40)
    assert tuple(res["results"].shape) == (2, 20, 39) and bool(torch.isfinite(res["results"]).all())
