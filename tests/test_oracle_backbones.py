"""CPU: the backbones of BASELINE configs 4 / 5 (no reference source exists: PARITY UNPINNED) -- structural pins of the
parameter tables and the oracle restatements against the published architecture facts."""
import numpy as np
import torch
import torch.nn.functional as F

import h3d_amd  # noqa: F401
from h3d_amd import arch_hg, model, synth
from oracle import hourglass as ohg

HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}


def _nparams(shapes):
    return sum(int(np.prod(s)) for k, s in shapes.items()
               if len(s) and not k.endswith(("running_mean", "running_var")))


def test_hourglass_table_matches_published_size_and_module_tree():
    body = _nparams(arch_hg.state_dict_shapes({}))
    assert abs(body - 187.7e6) < 0.5e6, body                       # Hourglass-104 body (CenterNet reports 191 M with ctdet heads)
    shapes = arch_hg.state_dict_shapes(HEADS)
    m = model.hourglass_net(HEADS)
    assert set(m.state_dict()) == set(shapes)
    assert all(tuple(v.shape) == tuple(shapes[k]) for k, v in m.state_dict().items())
    # innermost module: 4 residuals at 512 channels; the stride-2 residuals carry a 1x1 skip
    assert shapes["kps.0.low2.low2.low2.low2.low2.3.conv1.weight"] == (512, 512, 3, 3)
    assert shapes["kps.1.low1.0.skip.0.weight"] == (256, 256, 1, 1) and "kps.1.up1.0.skip.0.weight" not in shapes
    assert shapes["hm.1.0.conv.bias"] == (256,) and shapes["hps.0.1.weight"] == (34, 256, 1, 1)
    assert model.create_model("hourglass", HEADS).arch_name == "hourglass"
    assert model.create_model("dla_34", HEADS).arch_name == "dla34"
    assert abs(arch_hg.conv_flops(HEADS) / 1e9 - 701.0) < 1.0


def test_hourglass_oracle_shapes_and_identities():
    heads = {"hm": 1, "wh": 2}
    sd = synth.synth_state_dict(arch_hg.state_dict_shapes(heads), seed=0, gain=0.8)
    net = ohg.HourglassOracle(sd, heads)
    x = torch.from_numpy(synth.synth_images(1, 128, 256))
    with torch.no_grad():
        outs = net(x)
    assert len(outs) == 2 and outs[1]["hm"].shape == (1, 1, 32, 64) and outs[0]["wh"].shape == (1, 2, 32, 64)
    assert all(torch.isfinite(v).all() for o in outs for v in o.values())
    # the engine evaluates `nn.Upsample(scale_factor=2)` (nearest) with its depthwise-deconv + add kernel and a 0/1 tap
    # table (engine.PackedWeights.nearest_up_key): ConvTranspose2d(k=4, s=2, p=1) with taps (1..2, 1..2) = 1 IS nearest x2
    t = torch.from_numpy(synth.uniform("t", (2, 8, 5, 7), -1, 1))
    wn = torch.zeros(8, 1, 4, 4)
    wn[:, 0, 1:3, 1:3] = 1.0
    assert torch.equal(F.conv_transpose2d(t, wn, None, stride=2, padding=1, groups=8), F.interpolate(t, scale_factor=2, mode="nearest"))
    # the stem is a plain 7x7 stride-2 conv: im2col order k = c*49 + ky*7 + kx is weight.reshape(Cout, -1)
    w = torch.from_numpy(sd["pre.0.conv.weight"])
    cols = F.unfold(x, 7, padding=3, stride=2)                                   # [1, 147, L], same k order
    y = (w.reshape(128, 147) @ cols[0]).reshape(1, 128, 64, 128)
    np.testing.assert_allclose(y.numpy(), F.conv2d(x, w, None, 2, 3).numpy(), rtol=1e-4, atol=1e-4)
