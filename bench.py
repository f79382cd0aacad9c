#!/usr/bin/env python
"""Headline benchmark: images/s of the multi_pose hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--dtype bf16]

One "step" = one batch of synthetic 512x512 images (already resident in HBM) through
DLA-34+DCNv2 (+pose/shape heads) -> sigmoid+NMS+top-k decode -> per-detection SMPL/LBS meshes,
and for N>1 the single all-gather of the decoded detections (RCCL).  One process per GPU
(torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE); per-GPU batch is fixed (weak scaling).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  "roofline"     -- dominant kernel family (by device time) timed live with HIP events on the
                    launch stream: algorithmic FLOP per launch / mean launch duration vs the dense
                    bf16 MFMA peak of MI355X;
  "cpu_baseline" -- the oracle's torch-CPU restatement of the same graph timed on this box's host
                    cores on a bounded sample (a reported baseline, not the target).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import h3d_amd  # noqa: E402,F401
from h3d_amd import _lib, arch, synth  # noqa: E402
from h3d_amd.detector import MultiPoseDetector, Opt, gather_detections  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def op_flops(op):
    """Algorithmic FLOPs of one plan op (2 per MAC; conv/deconv only, SURVEY 8d convention)."""
    px = op.B * op.Ho * op.Wo
    if op.kind in (_lib.OP_CONV, _lib.OP_CONV_STREAM, _lib.OP_DCN, _lib.OP_STEM):
        return 2.0 * px * op.Cout * op.Cin * op.ksize * op.ksize
    if op.kind in (_lib.OP_DCN_FUSED, _lib.OP_DCN_FUSED_F16, _lib.OP_DCN_FUSED_STREAM):
        return 2.0 * px * (op.Cout + 27) * op.Cin * 9
    if op.kind == _lib.OP_UPDCN_F16:    # DeformConv (+ offset conv) at the output resolution, plus the 2x2-tap up-sampling
        return 2.0 * px * ((op.Cout + 27) * op.Cin * 9 + op.Cin * 4)
    if op.kind == _lib.OP_STEM3:        # 7x7 3->16 and 3x3 16->16 at full resolution, 3x3 16->32 at half
        return 2.0 * op.B * (op.H * op.W * 16 * (3 * 49 + 16 * 9) + op.Ho * op.Wo * 32 * 16 * 9)
    if op.kind == _lib.OP_UPADD:
        return 2.0 * px * op.Cout * 4
    if op.kind == _lib.OP_HEADS:
        d = ctypes.cast(op.in2, ctypes.POINTER(_lib.H3dHeadsDesc)).contents
        return sum(2.0 * px * op.Cout * (op.Cin * 9 + d.head[i].C) for i in range(d.nheads))
    return 0.0


def kernel_name(op):
    buf = ctypes.create_string_buffer(200)
    _lib.check(_lib.lib().h3d_op_kernel_name(ctypes.byref(op), buf, 200), "op_kernel_name")
    return buf.value.decode()


def per_kernel_profile(plan, iters):
    """HIP-event time of every op of the plan, grouped by kernel instantiation."""
    n = len(plan.ops)
    ms = (ctypes.c_float * n)()
    tot = np.zeros(n)
    for _ in range(iters):
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "run_ops_timed")
        tot += np.frombuffer(ms, dtype=np.float32, count=n)
    tot /= iters
    groups = {}
    for i, op in enumerate(plan.ops):
        real_cout = op.Cout
        g = groups.setdefault(kernel_name(op), {"ms": 0.0, "flops": 0.0, "launches": 0})
        g["ms"] += float(tot[i])
        g["flops"] += op_flops(op)
        g["launches"] += 1
    return groups


def cpu_baseline(opt, sd, seconds_budget=20.0):
    """Oracle (torch-CPU restatement, kind 'port') on a bounded sample of the same workload."""
    from oracle import decode as odec, dla as odla, smpl as osmpl
    from h3d_amd import smpl as psmpl
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                          # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    net = odla.DLAOracle(sd, opt.heads, use_dcn=not opt.not_use_dcn)
    model = psmpl.SMPLModel.synthetic().numpy_dict()
    B = 2
    x = torch.from_numpy(synth.synth_images(B, opt.input_h, opt.input_w))

    def step():
        with torch.no_grad():
            o = {k: v.numpy() for k, v in net(x)[0].items()}
        dets, aux = odec.multi_pose_decode(odec.sigmoid_clamp(o["hm"]), o["wh"], o["hps"], o["reg"],
                                           odec.sigmoid_clamp(o["hm_hp"]), o["hp_offset"], K=opt.K, return_aux=True)
        n = 4                                             # meshes per image in the CPU sample (fp64 numpy)
        idx = aux["inds"][:, :n]
        th = np.stack([o["pose"][i].reshape(72, -1)[:, idx[i]].T for i in range(B)]).reshape(-1, 72)
        be = np.stack([o["shape"][i].reshape(10, -1)[:, idx[i]].T for i in range(B)]).reshape(-1, 10)
        osmpl.lbs(be, th, model)

    t0 = time.time()
    step()                                                # warm-up
    print("[bench] cpu_baseline warm-up step %.1f s on %d threads" % (time.time() - t0, cores), file=sys.stderr, flush=True)
    t0 = time.time()
    iters = 0
    while iters < 2 or (time.time() - t0 < seconds_budget and iters < 12):
        step()
        iters += 1
        print("[bench] cpu_baseline step %d: %.1f s elapsed" % (iters, time.time() - t0), file=sys.stderr, flush=True)
    dt = time.time() - t0
    return {"value": round(B * iters / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d steps of batch %d (512x512, DLA-34+DCNv2+heads fp32 via torch-CPU threads=%d, numpy decode, "
                      "4 fp64 SMPL meshes/image)" % (iters, B, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--people", type=int, default=100, help="SMPL meshes per image (<= K)")
    ap.add_argument("--streams", type=int, default=1, help="sub-batches run concurrently on their own HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm

    opt = Opt(input_h=512, input_w=512, smpl=True, smpl_people=args.people, dtype=args.dtype, K=100)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
    det.model.engine(dev).streams = args.streams
    images = torch.from_numpy(synth.synth_images(1, 512, 512, seed=317 + rank)).to(dev)
    images = images.expand(args.batch, 3, 512, 512).contiguous()
    images += 0.01 * torch.arange(args.batch, device=dev, dtype=torch.float32).view(-1, 1, 1, 1)   # distinct images

    def step():
        res = det.run(images)
        return gather_detections(res["dets"]) if world > 1 else res["dets"]

    # one-time setup outside both warm-up and the timed region: weight packing + plan lowering (host work and uploads,
    # no network launches) and the SMPL model upload.  The first launch of every kernel still pays its code-object
    # load, so keep --warmup >= 1
    t_setup = time.perf_counter()
    eng = det.model.engine(dev)
    if args.streams <= 1:                       # (the sub-batch plans of --streams N are built by the first step)
        eng.plan(args.batch, 512, 512)
    from h3d_amd import smpl as _smpl
    if det.smpl_model._dev is None:
        det.smpl_model._dev = _smpl._device_pack(det.smpl_model, dev)
    torch.cuda.synchronize()
    t_built = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        print("[bench] setup (weight packing, plan) %.1f s; warm-up (%d steps) %.1f s" % (t_built - t_setup, args.warmup,
                                                                                         time.perf_counter() - t_built),
              file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == (world * args.batch, 100, 40)
    if rank == 0:
        print("[bench] %d GPU(s): %.1f images/s, %.3f ms/step" % (world, world * args.batch * args.steps / dt,
                                                                 1e3 * dt / args.steps), file=sys.stderr, flush=True)

    if rank == 0:
        gflop_img = arch.conv_flops(opt.heads, True) / 1e9
        line = {
            "metric": "images/sec whole-node, DLA-34+SMPL batch-64 512x512; top-k index bit-match",
            "value": round(world * args.batch * args.steps / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DLA-34+DCNv2 multi_pose + pose/shape heads -> sigmoid/NMS/top-100 decode -> "
                                   "SMPL 6890-vert LBS; 512x512, batch %d per GPU (BASELINE configs[2])" % args.batch,
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "K": 100,
                       "smpl_people_per_image": args.people, "conv_gflop_per_image": round(gflop_img, 2),
                       "parallelism": "dp%d (image shards, one all-gather of dets)" % world,
                       "weights": "synthetic (h3d_amd.synth, seed 0)"},
        }
        line["model_tflops"] = round(gflop_img * line["value"] / 1e3, 1)
        if not args.no_roofline:
            plan = det.model.engine(dev).plan(args.batch, 512, 512)
            groups = per_kernel_profile(plan, iters=3)
            total_ms = sum(g["ms"] for g in groups.values())
            name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            traffic = None          # HBM bytes per launch from the committed rocprofv3 --pmc passes of this build
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
                traffic = pmc.get(name, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError, KeyError):
                pass
            line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1),
                                "peak": PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else 157.3, "unit": "TFLOP/s",
                                "frac": round(ach / (PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else 157.3), 4),
                                "traffic": traffic, "traffic_source": "profiles/r01_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, "
                                "separate --pmc passes)" if traffic else None, "launches_per_step": g["launches"],
                                "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                                "share_of_network_time": round(g["ms"] / total_ms, 3),
                                "network_ms_per_step": round(total_ms, 3)}
            print("[bench] roofline %s" % json.dumps(line["roofline"]), file=sys.stderr, flush=True)
            line["kernels"] = {k: {"ms": round(v["ms"], 3), "n": v["launches"],
                                   "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                               for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}
            print("[bench] kernels %s" % json.dumps(line["kernels"]), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(opt, sd)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
