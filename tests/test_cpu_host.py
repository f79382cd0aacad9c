"""CPU (-m "not gpu"): host logic, the C-ABI library's exports, sharding + the world_size-2 gloo
path of the one collective.  No compute calls (there is no GPU here and no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import h3d_amd  # noqa: F401
from h3d_amd import _lib, arch, detector, model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "h3d.h")).read()
    declared = set(re.findall(r"\b(h3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), name
    size_t_fns = {"h3d_dcn_v2_workspace_bytes", "h3d_dcn_v2_packed_weight_bytes", "h3d_dcn_v2_packed_workspace_bytes",
                  "h3d_nms_topk_large_workspace_bytes"}     # (size_t results: bound separately)
    assert declared - {"h3d_last_error", "h3d_abi_version"} - size_t_fns == set(_lib.SIGNATURES)
    assert L.h3d_dcn_v2_workspace_bytes(2, 64, 10, 12, 64) >= 2 * 10 * 12 * (64 + 32) * 4 + 128 * 9 * 64 * 4
    assert L.h3d_dcn_v2_workspace_bytes(0, 64, 10, 12, 64) == 0
    assert L.h3d_abi_version() == _lib.ABI_VERSION


def test_op_struct_matches_header_layout():
    # 2 int32, 5 pointers, 18 int32 (ABI 2: + wexp, wexp2) -> 8 + 40 + 72 = 120 bytes on LP64
    assert ctypes.sizeof(_lib.H3dOp) == 120 and _lib.H3dOp.wexp.offset == 112
    assert _lib.H3dOp.in_.offset == 8 and _lib.H3dOp.B.offset == 48


def test_null_and_bad_arguments_return_error_codes_without_a_gpu():
    L = _lib.lib()
    rc = L.h3d_run_ops(None, 0, None)
    assert rc == -5 and b"null plan" in L.h3d_last_error()
    rc = L.h3d_nms_topk(None, 1, 1, 4, 4, 2, 0, None, None, None, None, None)
    assert rc == -5
    with pytest.raises(RuntimeError, match="argument error"):
        _lib.check(rc, "nms_topk")


def test_product_has_no_cpu_fallback():
    m = model.dla_net(HEADS, not_use_dcn=True)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        m(torch.zeros(1, 3, 64, 64))
    from h3d_amd import decode
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        decode.multi_pose_decode(torch.zeros(1, 1, 8, 8), torch.zeros(1, 2, 8, 8), torch.zeros(1, 34, 8, 8), K=4)
    src = "".join(open(os.path.join(ROOT, "human-3d-reconstruction_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "human-3d-reconstruction_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_flop_table_matches_survey():
    # 80.48 / 75.83 GFLOP per image incl. two dead 1x1 `project` convs (0.134 GFLOP) we skip
    assert abs(arch.conv_flops(HEADS, True) / 1e9 - (80.48 - 0.134)) < 0.02
    assert abs(arch.conv_flops(HEADS, False) / 1e9 - (75.83 - 0.134)) < 0.02


def test_default_init_follows_reference_rules():
    torch.manual_seed(0)
    m = model.dla_net(HEADS)
    sd = m.state_dict()
    assert float(sd["hm.2.bias"]) == pytest.approx(-2.19) and float(sd["hm_hp.2.bias"][3]) == pytest.approx(-2.19)
    assert sd["wh.0.bias"].abs().max() == 0 and sd["reg.2.bias"].abs().max() == 0
    assert sd["hm.0.bias"].abs().max() > 0
    assert sd["dla_up.ida_0.proj_1.conv.conv_offset_mask.weight"].abs().max() == 0
    assert sd["dla_up.ida_0.proj_1.conv.bias"].abs().max() == 0
    w = sd["ida_up.up_2.weight"]
    assert w.shape == (64, 1, 8, 8) and torch.equal(w[0], w[5])
    assert float(w[0, 0, 3, 3]) == pytest.approx((1 - abs(3 / 4 - 0.875)) ** 2)
    assert float(sd["base.level2.root.bn.running_var"][0]) == 1.0


def test_shard_batch_partitions():
    for n, world in [(64, 8), (10, 4), (3, 8), (128, 8)]:
        spans = [detector.shard_batch(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.parametrize("world,port", [(2, 29531), (4, 29537)])
def test_gather_detections_gloo(world, port):
    # even and uneven shards (n = 8, 9, 10, ...), see tests/dist_worker.py
    script = os.path.join(ROOT, "tests", "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(port), script],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout


def test_bench_launches_its_own_ranks_gloo_dry_run():
    # `python bench.py --gpus 2` without torchrun must start 2 ranks itself and relay rank 0's JSON line; --dry-run
    # replaces the GPU step by the collective on CPU tensors over gloo (no HIP call anywhere), everything else --
    # launcher, rendezvous on 127.0.0.1, barriers, max-over-ranks timing, the line's contract fields -- is the real code
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run", "--batch", "5"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 10 and line["dry_run"] is True and line["scaling"] == "weak"


def test_bench_strong_scaling_mode_splits_one_batch_gloo_dry_run():
    # `--global-batch G`: ONE batch of G images per step for the whole job, rank r takes shard_batch(G, r, world) -- the
    # reference's DataParallel scatter (trains/trainer.py:176; SURVEY 8e: 64 -> 8 per GPU).  Uneven split (9 over 2 ranks: 5 + 4):
    # the padded fixed-size all-gather and the pad-row removal are the real code; the line says "strong".
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run", "--global-batch", "9"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert line["config"]["global_batch"] == 9 and line["config"]["batch_per_gpu"] == 5       # rank 0's share


def test_flip_helpers_match_the_reference_semantics():
    # models/utils.py:29-51 (flip test): host helpers (torch, device-resident) vs the numpy restatement
    import numpy as np
    import torch
    from h3d_amd import utils
    from oracle import decode as odec
    flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]      # COCO left/right joints
    rng = np.random.default_rng(0)
    hm = rng.standard_normal((2, 17, 6, 10)).astype(np.float32)
    hps = rng.standard_normal((2, 34, 6, 10)).astype(np.float32)
    assert np.array_equal(utils.flip_tensor(torch.from_numpy(hm)).numpy(), odec.flip_tensor(hm))
    assert np.array_equal(utils.flip_lr(torch.from_numpy(hm), flip_idx).numpy(), odec.flip_lr(hm, flip_idx))
    assert np.array_equal(utils.flip_lr_off(torch.from_numpy(hps), flip_idx).numpy(), odec.flip_lr_off(hps, flip_idx))
    # overlapping pairs are applied in sequence, as the reference's loop does
    odd = [[0, 1], [1, 2]]
    assert np.array_equal(utils.flip_lr(torch.from_numpy(hm), odd).numpy(), odec.flip_lr(hm, odd))


def test_f16x3_filter_split_is_exact_to_23_bits_and_prescaled_into_the_normal_range():
    """Host half of the f16x3 plans (engine.x3_split / x3_exp; csrc/common.h ET<x3_t>): per 8 consecutive K elements the 8 fp16 high
    terms then the 8 low terms in the 32 bytes of the fp32 values; hi + lo reproduces x to 2^-23 relative once the bank is pre-scaled
    so that its low terms are normal fp16 numbers (an unscaled 0.05 is only good to 2^-20.7)."""
    import numpy as np
    from h3d_amd import engine
    rng = np.random.default_rng(0)
    w = torch.from_numpy((rng.standard_normal((24, 9, 32)) * 0.05).astype(np.float32))
    e = engine.x3_exp(w)
    assert 2.0 ** 13 <= float(w.abs().max()) * 2.0 ** e < 2.0 ** 14
    s = engine.x3_split(w * 2.0 ** e)
    assert s.shape == w.shape and s.dtype == torch.float32
    h = s.contiguous().view(torch.float16).reshape(-1, 4, 2, 8)          # [rows, K/8, (hi | lo), 8]
    hi, lo = h[:, :, 0].float().reshape(w.shape), h[:, :, 1].float().reshape(w.shape)
    assert torch.equal(hi, (w * 2.0 ** e).half().float())
    back = (hi.double() + lo.double()) * 2.0 ** -e
    rel = ((back - w.double()).abs() / w.double().abs().clamp_min(1e-30))
    big = w.abs() > float(w.abs().max()) * 2.0 ** -15                    # filters whose low term is a normal fp16 number
    assert float(rel[big].max()) <= 2.0 ** -23 * 1.01
    # without the pre-scale the same bank is an order of magnitude coarser: the reason h3d_op.wexp exists
    s0 = engine.x3_split(w).contiguous().view(torch.float16).reshape(-1, 4, 2, 8)
    back0 = s0[:, :, 0].double().reshape(w.shape) + s0[:, :, 1].double().reshape(w.shape)
    assert float(((back0 - w.double()).abs() / w.double().abs().clamp_min(1e-30))[big].max()) > 2.0 ** -21
    assert engine.x3_exp(torch.zeros(4, 8)) == 0


def test_build_flags_and_default_library_has_no_superseded_generations():
    L = _lib.lib()
    flags = L.h3d_build_flags()
    assert flags & ~3 == 0
    assert _lib.has_extra() == bool(flags & 1)


def test_plan_faithful_emulation_rounds_where_the_plans_round():
    """oracle/dla.py emulate='bf16_plan' (round 5): small input, CPU only -- it is a bf16-level evaluation of the graph
    (close to the fp32 oracle, not equal), differs from the older conv-input emulation, and with no rounding point selected
    (plan_parts=[]) it is that older emulation's treatment up to the up-sampling tap weights, which the plans keep in fp32."""
    import numpy as np
    from h3d_amd import synth
    from oracle import dla as odla
    heads = {"hm": 1, "wh": 2}
    sd = synth.synth_state_dict(arch.state_dict_shapes(heads, True), seed=0, gain=1.25)
    x = torch.from_numpy(synth.synth_images(1, 64, 64, seed=3))
    with torch.no_grad():
        ref = odla.DLAOracle(sd, heads, use_dcn=True)(x)[0]["hm"]
        old = odla.DLAOracle(sd, heads, use_dcn=True, emulate="bf16")(x)[0]["hm"]
        plan = odla.DLAOracle(sd, heads, use_dcn=True, emulate="bf16_plan")(x)[0]["hm"]
        none = odla.DLAOracle(sd, heads, use_dcn=True, emulate="bf16_plan", plan_parts=[])(x)[0]["hm"]
    scale = float(ref.abs().max())
    for t in (old, plan, none):
        e = float((t - ref).abs().max())
        assert 1e-4 * scale < e < 0.2 * scale, (e, scale)
    assert float((plan - old).abs().max()) > 0
    # plan_parts=[]: conv inputs and raw filters rounded, BatchNorm in fp32 -- the 'bf16' mode's arithmetic except for the depthwise
    # up-sampling weights (fp32 in every plan, rounded by emulate='bf16'): closer to it than either is to fp32
    assert float((none - old).abs().max()) < 0.5 * float((old - ref).abs().max())
