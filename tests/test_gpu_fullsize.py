"""BASELINE configs[1] (batch 32) and configs[2] (batch 64) at their real size: 512x512 bf16, the plan bench.py times.

Two distinct synthetic images repeated B/2 times go through the default bf16 plan (the same kernels, tile
configurations and grid sizes as the benchmark); then
  * the heads of the two images are compared with the fp32 oracle network (oracle/dla.py, pinned to the reference's
    own model output by tests/test_oracle_golden.py).  The tolerance is not a constant: the synthetic weights with
    gain 1.25 keep the signal alive through DLA-34 (head maps with std 1-2.5), which also amplifies rounding noise,
    so the yardstick is a CPU evaluation of the same graph that rounds WHERE THE PLAN ROUNDS (DLAOracle(emulate="bf16_plan"),
    round 5: BatchNorm folded into the filters before they are rounded, every stored activation rounded -- residual and skip
    operands included --, fp16 DeformConv filters / samples / blend, fp16 node inputs).  Per head, the GPU's distance from
    the fp32 result may be at most 1.5x that evaluation's in max-norm and in rms and 1.25x in the 99.99 % quantile of |error|.
    (Rounds 3-4 compared with emulate="bf16", which rounds conv inputs and raw filters only: it has a lighter error tail than
    any plan that stores 2-byte activations -- `hm`: max 0.38 / q99.99 0.36 against 0.52 / 0.47 for the plan's rounding points and
    0.52-0.62 / 0.46-0.57 on the GPU -- and in round 4 the max-norm bound was widened to the maximum over all heads to let that
    pass; tools/hm_tail.py + DESIGN.md section 9.2 have the bisection: stored residual / skip operands make the tail, the folded
    filters make the GPU's error correlate with the emulation's (0.20 -> 0.66) and reproduce its mean shift on `hm`.);
  * every repeat must be BIT-identical to the first occurrence, and a second forward to the first forward -- a race in
    a DMA ring / counted wait of the 8- and 16-wave variants shows up here;
  * the decoded top-k peak indices are compared with the oracle's (oracle/index_match.py): bit-identical on the
    ranks whose order the bf16 error cannot change, statistics asserted for the rest.
"""
import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from oracle import dla as odla
from oracle import index_match as oim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_RATIO = 1.5  # GPU bf16 error / CPU bf16-emulation error (both vs the fp32 oracle), max-norm and rms
Q_RATIO = 1.25    # ... and in the 99.99 % quantile of |error| (per head, against the plan-faithful emulation)
GAIN = 1.25       # signal-preserving synthetic weights (h3d_amd.synth): head maps with O(1) variation and separated peaks


def _err_stats(got, ref):
    """(max, rms, 99.99 % quantile) of |got - ref|"""
    e = np.abs(got - ref).ravel()
    return float(e.max()), float(np.sqrt(np.mean(e ** 2))), float(np.quantile(e, 0.9999))


@pytest.fixture(scope="module")
def setup():
    opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16", K=100)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=GAIN)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    two = synth.synth_images(2, 512, 512, seed=317)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        ref = {k: v.numpy() for k, v in odla.DLAOracle(sd, opt.heads, use_dcn=True)(torch.from_numpy(two))[0].items()}
        emu = {k: v.numpy() for k, v in odla.DLAOracle(sd, opt.heads, use_dcn=True, emulate="bf16_plan")(torch.from_numpy(two))[0].items()}
    tol = {k: _err_stats(emu[k], ref[k]) for k in ref}
    return opt, det, two, ref, tol, sd


@pytest.mark.parametrize("batch", [8, 16, 32, 64])   # 8 / 16 = the per-GPU shards of the headline batch on 8 / 4 GPUs (64 -> 8 per GPU)
def test_full_size_bf16_plan_vs_oracle(setup, batch):
    opt, det, two, ref, tol, _ = setup
    xs = torch.from_numpy(two).to(DEV).repeat(batch // 2, 1, 1, 1).contiguous()
    res = det.run(xs)
    heads = {k: v.clone() for k, v in res["heads"].items()}
    inds = res["inds"].clone()
    dets = res["dets"].clone()
    # (1) parity of the first two images
    got = {k: v[:2].cpu().numpy() for k, v in heads.items()}
    worst = {k: _err_stats(got[k], ref[k]) for k in opt.heads}
    print("batch %d: head error vs fp32 oracle (max, rms, q99.99): %s" % (batch, {k: tuple(round(a, 4) for a in t) for k, t in worst.items()}))
    print("          CPU emulation of the plan  (max, rms, q99.99): %s" % {k: tuple(round(a, 4) for a in t) for k, t in tol.items()})
    print("          ratios                     (max, rms, q99.99): %s" % {k: tuple(round(a / b, 3) for a, b in zip(worst[k], tol[k])) for k in worst})
    # every bound is per head, against the emulation's figure of THAT head
    for k, (emax, erms, eq) in worst.items():
        tmax, trms, tq = tol[k]
        assert emax <= BF16_RATIO * tmax + 1e-3 and erms <= BF16_RATIO * trms + 1e-4 and eq <= Q_RATIO * tq + 1e-3, (k, worst[k], tol[k])
    # (2) every repeat bit-identical, run-to-run bit-identical
    for k, v in heads.items():
        r = v.view(batch // 2, 2, *v.shape[1:])
        assert torch.equal(r, r[:1].expand_as(r)), "head %s differs between repeats of the same image" % k
    assert torch.equal(inds.view(batch // 2, 2, -1), inds[:2].unsqueeze(0).expand(batch // 2, 2, -1))
    again = det.run(xs)
    for k in heads:
        assert torch.equal(heads[k], again["heads"][k]), k
    assert torch.equal(dets, again["dets"])
    # (3) top-k peak indices vs the oracle
    m = oim.index_match(got, inds[:2].cpu().numpy(), ref, K=opt.K)
    print("batch %d: index_match %s" % (batch, m))
    assert m["robust_prefix_equal"], m
    # random-weight heat maps are noise-like: the median score gap between consecutive peaks (6e-4) is far below the
    # bf16 score error (0.07), so ranks shuffle; the peaks themselves mostly survive (measured overlap 0.70)
    assert m["set_overlap"] >= 0.5, m
    assert res["verts"].shape == (batch, opt.K, 6890, 3)
    del res, again, heads
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_full_size_f32_mode_indices_match_oracle(setup, dtype):
    """Parity mode (fp32 activations, exact fmaf chains on v_mfma_f32_32x32x2_f32) at 512x512: here the metric's
    "top-k index bit-match" is attainable end to end -- heads within 2e-3 of the oracle (max |head| ~ 10) and the decoded
    peak indices identical to the oracle's wherever the score gap exceeds that error.  "f16x3" (round 5) is the same contract on
    the fp16 matrix cores: fp32 storage, three fp16 MFMAs on split operands per fp32 product."""
    opt, _, two, ref, _, sd = setup
    o32 = Opt(input_h=512, input_w=512, smpl=True, dtype=dtype, K=100)
    det = MultiPoseDetector(o32, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    res = det.run(torch.from_numpy(two).to(DEV))
    got = {k: v.cpu().numpy() for k, v in res["heads"].items()}
    for k in o32.heads:
        e = float(np.abs(got[k] - ref[k]).max())
        assert e <= 2e-3, (k, e)
    m = oim.index_match(got, res["inds"].cpu().numpy(), ref, K=100)
    print("%s mode: max head error %.3g, index_match %s" % (dtype, max(float(np.abs(got[k] - ref[k]).max()) for k in o32.heads), m))
    assert m["robust_prefix_equal"] and m["agreement"] >= (0.99 if dtype == "f16x3" else 0.95) and m["set_overlap"] >= 0.98, m
    # ... and the bit-match clause is not vacuous here: some ranks are provably stable under the measured score error
    # (bench.py's parity_mode record on the same weights: robust_prefix 5, equal_prefix 100)
    assert m["robust_prefix"] > 0, m
    # the metric's third clause end to end ("vertices within 1e-4"): the product's meshes against the CPU path's -- the fp32 oracle's pose /
    # shape maps read at the product's own centre indices -> oracle/smpl.py in fp64 (bench.py parity_mode reports the same figure)
    from oracle import smpl as osmpl
    n = res["verts"].shape[1]
    inds = res["inds"].cpu().numpy()[:, :n]
    th = np.concatenate([ref["pose"][i].reshape(72, -1)[:, inds[i]].T for i in range(inds.shape[0])])
    be = np.concatenate([ref["shape"][i].reshape(10, -1)[:, inds[i]].T for i in range(inds.shape[0])])
    v_ref, _ = osmpl.lbs(be, th, det.smpl_model.numpy_dict())
    e = float(np.abs(res["verts"].cpu().numpy().reshape(-1, v_ref.shape[1], 3) - v_ref).max())
    print("%s mode: meshes of %d detections, max |vertex - CPU path's| %.3g" % (dtype, v_ref.shape[0], e))
    assert e <= 1e-4, e


def _lowp_vs_emulation(got, ref, emu, what, ratio=BF16_RATIO):
    # the CPU emulation rounds where the plan rounds (conv weights and every stored activation, including the skip /
    # down-sample conv outputs in front of the residual adds: oracle/hourglass.py, oracle/resdcn.py), so the same 1.5x
    # as for DLA-34 applies, in max-norm and in rms (round 2 had 3x / 2x here against an emulation that rounded conv
    # inputs only; f32 mode pins the wiring at 2e-3 in tests/test_gpu_backbones.py)
    emax, erms = float(np.abs(got - ref).max()), float(np.sqrt(np.mean((got - ref) ** 2)))
    tmax, trms = float(np.abs(emu - ref).max()), float(np.sqrt(np.mean((emu - ref) ** 2)))
    print("%s: GPU error (max %.4g, rms %.4g) vs CPU emulation of the same arithmetic (max %.4g, rms %.4g); head scale %.3g"
          % (what, emax, erms, tmax, trms, float(np.abs(ref).max())))
    assert emax <= ratio * tmax + 1e-3 and erms <= ratio * trms + 1e-4, (what, emax, erms, tmax, trms)


def test_full_size_hourglass_shard_vs_oracle():
    """BASELINE configs[3]: Hourglass-104 multi_pose, batch 128 over 8 GPUs = 16 images of 512x512 per GPU, bf16.  One
    synthetic image repeated 16 times: heads of both stacks vs the oracle restatement (parity unpinned: no reference
    source), every repeat bit-identical, decode bit-exact on the GPU's own heads."""
    from h3d_amd import arch_hg, decode, utils
    from h3d_amd.detector import make_detector
    from oracle import decode as odec
    from oracle import hourglass as ohg
    opt = Opt(arch="hourglass", input_h=512, input_w=512, dtype="bf16", K=100)
    sd = synth.synth_state_dict(arch_hg.state_dict_shapes(opt.heads), seed=0, gain=0.8)
    det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    one = synth.synth_images(1, 512, 512, seed=317)
    with torch.no_grad():
        ref = ohg.HourglassOracle(sd, opt.heads)(torch.from_numpy(one))
        emu = ohg.HourglassOracle(sd, opt.heads, emulate_bf16=True)(torch.from_numpy(one))
    xs = torch.from_numpy(one).to(DEV).repeat(16, 1, 1, 1).contiguous()
    outs = [{k: v.clone() for k, v in o.items()} for o in det.model(xs)]
    for i, (o, r, e) in enumerate(zip(outs, ref, emu)):
        for k in opt.heads:
            assert torch.equal(o[k], o[k][:1].expand_as(o[k])), (i, k)
            _lowp_vs_emulation(o[k][:1].cpu().numpy(), r[k].numpy(), e[k].numpy(), "hourglass stack %d %s" % (i, k))
    res = det.run(xs)
    h = {k: v[:1].cpu().numpy() for k, v in res["heads"].items()}
    dref = odec.multi_pose_decode(utils._sigmoid(res["heads"]["hm"][:1].clone()).cpu().numpy(), h["wh"], h["hps"], h["reg"],
                                  utils._sigmoid(res["heads"]["hm_hp"][:1].clone()).cpu().numpy(), h["hp_offset"], K=100)
    np.testing.assert_array_equal(res["dets"][:1].cpu().numpy(), dref)
    assert torch.equal(res["dets"], res["dets"][:1].expand_as(res["dets"]))
    del det, outs, res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_full_size_resdcn_shard_vs_oracle(dtype):
    """BASELINE configs[4]: ResNet-101-DCN ctdet at 768x768 fp16, batch 256 over 8 GPUs = 32 images per GPU -- in the
    config's own arithmetic (dtype f16: H3D_F16 plans) and in bf16.  One synthetic image repeated 32 times vs the oracle
    restatement (parity unpinned) and its emulation of the same arithmetic."""
    from h3d_amd import arch_res
    from h3d_amd.detector import make_detector
    from oracle import resdcn as ores
    opt = Opt(arch="resdcn_101", task="ctdet", input_h=768, input_w=768, dtype=dtype, K=100)
    sd = synth.synth_state_dict(arch_res.state_dict_shapes(opt.heads, 64), seed=0, gain=0.9, offset_scale=1.0)
    det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
    one = synth.synth_images(1, 768, 768, seed=317)
    with torch.no_grad():
        ref = ores.ResDCNOracle(sd, opt.heads)(torch.from_numpy(one))[0]
        emu = ores.ResDCNOracle(sd, opt.heads, emulate=dtype)(torch.from_numpy(one))[0]
    xs = torch.from_numpy(one).to(DEV).repeat(32, 1, 1, 1).contiguous()
    res = det.run(xs)
    if dtype == "f16":
        from gpu_helpers import kernel_name
        names = {kernel_name(op) for op in det.model.engine(torch.device(DEV)).plan(32, 768, 768).ops}
        assert all("f16_t" in n or n.startswith("dcn5_kernel<") for n in names), sorted(n for n in names if "f16_t" not in n)
    for k in opt.heads:
        v = res["heads"][k]
        assert torch.equal(v, v[:1].expand_as(v)), k
        _lowp_vs_emulation(v[:1].cpu().numpy(), ref[k].numpy(), emu[k].numpy(), "resdcn %s %s" % (dtype, k))
    assert res["dets"].shape == (32, 100, 6) and torch.equal(res["dets"], res["dets"][:1].expand_as(res["dets"]))
    del det, res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_end_to_end_f32_plain_plan_vs_reference_output(golden_dir, dtype):
    """The metric's second clause ("bit-exact top-k peak indices vs the reference's own CPU path") against reference
    OUTPUT, images -> indices: tests/golden/e2e_plain_512.npz is what the imported reference returns for
    `dla_net(heads, not_use_dcn=True)` (model.py:501-516) -> `_sigmoid` (utils.py:8-10; trainer.py:93,127) ->
    `multi_pose_decode(K=100)` (decode.py:77-163; trainer.py:456-469) on synth_images(2, 512, 512, seed=317) with
    gain-1.25 weights (oracle/gen_golden.py gen_e2e).  The product runs the SAME call order through its reference-named
    entry points (h3d_amd.model.dla_net -> utils._sigmoid -> decode.multi_pose_decode), f32 plan, plain convs.
    Tolerances: logits within 2e-3 abs (|logit| <= 40: 5e-5 relative; fp32 summation order + folded BatchNorm);
    peak indices bit-identical on every rank whose order cannot change under the measured score difference (asserted
    non-empty), positional agreement >= 0.95; detection rows of agreeing ranks equal within 2e-3 px."""
    import os
    from h3d_amd import decode, model, utils
    from oracle import decode as odec
    g = np.load(os.path.join(golden_dir, "e2e_plain_512.npz"))
    heads = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}
    sd = synth.synth_state_dict(arch.state_dict_shapes(heads, False), seed=0, gain=GAIN)
    m = model.dla_net(heads, not_use_dcn=True, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    out = m(torch.from_numpy(synth.synth_images(2, 512, 512, seed=317)).to(DEV))[0]
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for k in ("hm", "hm_hp"):
        e = float(np.abs(got[k] - g[k]).max())
        print("e2e vs reference output: %s max abs logit error %.3g (scale %.3g)" % (k, e, float(np.abs(g[k]).max())))
        assert e <= 2e-3, (k, e)
    for k in ("wh", "hps", "reg", "hp_offset"):
        e = float(np.abs(got[k][:, :, ::4, ::4] - g[k + "_s4"]).max())
        assert e <= 2e-3, (k, e)
    # the reference's call order on the product's tensors
    hm = utils._sigmoid(out["hm"].clone())
    hm_hp = utils._sigmoid(out["hm_hp"].clone())
    dets = decode.multi_pose_decode(hm, out["wh"], out["hps"], reg=out["reg"], hm_hp=hm_hp, hp_offset=out["hp_offset"], K=100)
    s, inds, clses, ys, xs = decode._topk(decode._nms(hm), K=100)
    inds = inds.cpu().numpy()
    np.testing.assert_allclose(hm.cpu().numpy(), g["hm_sig"], rtol=0, atol=5e-4)
    m_ = oim.index_match(got, inds, {"hm": g["hm"]}, K=100)
    print("e2e vs reference output: index_match %s" % m_)
    # ranks whose order no logit error of the measured size can change (oracle/index_match.robust_prefix_logit: the analysis in
    # logit space -- near a score of 1 the sigmoid compresses a 0.0066 logit gap to 7e-6, below the map-wide score error):
    # 6 and 12 on these two images; they must be bit-identical, and the set must not be empty
    assert m_["robust_prefix_logit"] > 0 and m_["robust_prefix_logit_equal"] and m_["robust_prefix_equal"], m_
    assert m_["agreement"] >= 0.95 and m_["set_overlap"] >= 0.98, m_      # (measured on the round-4 box: all 100 ranks of both images identical)
    same = inds == g["topk_inds"]
    np.testing.assert_allclose(s.cpu().numpy()[same], g["topk_scores"][same], rtol=0, atol=5e-4)
    d = dets.cpu().numpy()
    np.testing.assert_allclose(d[same][:, :5], g["dets"][same][:, :5], rtol=0, atol=2e-3)
    np.testing.assert_array_equal(d[same][:, 39], g["dets"][same][:, 39])
    # the product's decode on the REFERENCE's post-sigmoid maps: indices bit-identical on every rank (no ties among real peaks)
    s2, i2, _, _, _ = decode._topk(decode._nms(torch.from_numpy(g["hm_sig"]).to(DEV)), K=100)
    np.testing.assert_array_equal(i2.cpu().numpy(), g["topk_inds"])
    np.testing.assert_array_equal(s2.cpu().numpy(), g["topk_scores"])
    # joint heat maps: indices on each channel's tie-free prefix (several maps saturate at the 1 - 1e-4 clamp)
    hs, hi, _, _ = decode._topk_channel(decode._nms(hm_hp), K=100)
    hi = hi.cpu().numpy()
    checked = 0
    for b in range(2):
        for j in range(17):
            q = odec.strict_prefix(g["hp_scores"][b, j], 100)
            checked += q
            assert q == 0 or (hi[b, j, :q] == g["hp_inds"][b, j, :q]).mean() >= 0.95, (b, j, q)
    assert checked >= 500
