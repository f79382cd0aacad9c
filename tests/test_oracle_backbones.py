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


def test_resdcn_table_and_oracle():
    from h3d_amd import arch_res
    from oracle import resdcn as ores
    heads = {"hm": 80, "wh": 2, "reg": 2}
    shapes = arch_res.state_dict_shapes(heads)
    assert abs(_nparams(arch_res.state_dict_shapes({})) - 49.5e6) < 0.6e6             # ResNet-101 trunk 42.5 M + DCN / deconv stages
    m = model.resdcn_net(heads)
    assert set(m.state_dict()) == set(shapes) and m.arch_name == "resdcn101"
    assert shapes["layer3.22.conv2.weight"] == (256, 256, 3, 3) and shapes["layer1.0.downsample.0.weight"] == (256, 64, 1, 1)
    assert shapes["deconv_layers.0.conv_offset_mask.weight"] == (27, 2048, 3, 3) and shapes["deconv_layers.15.weight"] == (64, 64, 4, 4)
    assert model.create_model("resdcn_101", heads).head_conv == 64
    sd = synth.synth_state_dict(shapes, seed=0, gain=0.9)
    x = torch.from_numpy(synth.synth_images(1, 64, 96))
    with torch.no_grad():
        out = ores.ResDCNOracle(sd, heads)(x)[0]
    assert out["hm"].shape == (1, 80, 16, 24) and all(torch.isfinite(v).all() for v in out.values())


def test_transposed_conv_as_conv3_plus_depth2space_identity():
    # engine.PackedWeights.deconv4_as_conv3: ConvTranspose2d(C, C, 4, 2, 1) + BN == conv3x3 with 4C outputs + pixel shuffle
    from h3d_amd import engine
    C = 8
    w = torch.from_numpy(synth.uniform("wt", (C, C, 4, 4), -1, 1))
    bn = {"bn.weight": torch.from_numpy(synth.uniform("g", (C,), 0.5, 1.5)), "bn.bias": torch.from_numpy(synth.uniform("b", (C,), -1, 1)),
          "bn.running_mean": torch.from_numpy(synth.uniform("m", (C,), -1, 1)), "bn.running_var": torch.from_numpy(synth.uniform("v", (C,), 0.5, 2))}
    pw = object.__new__(engine.PackedWeights)
    pw.sd = dict(bn, **{"up.weight": w})
    wkey, bkey = pw.deconv4_as_conv3("up.weight", "bn")
    x = torch.from_numpy(synth.uniform("x", (2, C, 5, 7), -1, 1))
    ref = F.batch_norm(F.conv_transpose2d(x, w, None, stride=2, padding=1), bn["bn.running_mean"], bn["bn.running_var"],
                       bn["bn.weight"], bn["bn.bias"], False, 0.0, 1e-5)
    y = F.batch_norm(F.conv2d(x, pw.sd[wkey], None, 1, 1), pw.sd[bkey + ".running_mean"], pw.sd[bkey + ".running_var"],
                     pw.sd[bkey + ".weight"], pw.sd[bkey + ".bias"], False, 0.0, 1e-5)
    y = y.reshape(2, 2, 2, C, 5, 7).permute(0, 3, 4, 1, 5, 2).reshape(2, C, 10, 14)             # depth2space, group g = 2 py + px
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
