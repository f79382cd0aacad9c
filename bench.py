#!/usr/bin/env python
"""Headline benchmark: images/s of the multi_pose hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--dtype bf16]

One "step" = one batch of synthetic 512x512 images (already resident in HBM) through
DLA-34+DCNv2 (+pose/shape heads) -> sigmoid+NMS+top-k decode -> per-detection SMPL/LBS meshes,
and for N>1 the single all-gather of the decoded detections (RCCL).  One process per GPU
(torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE); per-GPU batch is fixed (weak scaling).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, spawned before anything touches the GPU) and relays
rank 0's JSON line; under torchrun it is a rank.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  "roofline"     -- dominant kernel family (by device time) timed live with HIP events on the
                    launch stream: algorithmic FLOP per launch / mean launch duration vs the dense
                    bf16 MFMA peak of MI355X;
                    plus network_frac / step_frac = the whole network's / whole step's FLOP rate vs that peak;
  "cpu_baseline" -- the oracle's torch-CPU restatement of the same graph timed on this box's host
                    cores on a bounded sample (a reported baseline, not the target);
  "index_match"  -- top-k peak indices of the GPU (bf16) path vs the fp32 oracle on the same 2 images
                    (oracle/index_match.py; the oracle is the checker here, never the thing measured).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

# more hardware queues than HIP's default of 4: with three or four steps in flight (small per-GPU shards) two of the streams
# otherwise share a queue and serialise (batch 8, four steps in flight: 5530 images/s with 4 queues, 7090 with 8).  A process
# setting read by the HIP runtime at start-up; must be in the environment before torch initialises HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import h3d_amd  # noqa: E402,F401
from h3d_amd import _lib, arch, synth  # noqa: E402
from h3d_amd.detector import MultiPoseDetector, Opt, gather_detections, shard_batch  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X dense bf16 / fp16 (MI355X_MICROARCH.md, chip-level parameters)
# f16x3: every algorithmic FLOP costs three fp16 MFMA FLOP (hi.hi + hi.lo + lo.hi), so the plan's ceiling on ALGORITHMIC FLOP is a third
PEAK_MFMA_TFLOPS = {"bf16": PEAK_BF16_MFMA_TFLOPS, "f16": PEAK_BF16_MFMA_TFLOPS, "f32": 157.3, "f16x3": round(PEAK_BF16_MFMA_TFLOPS / 3, 1)}
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC_FILE = "r05_pmc_traffic.json"   # this round's counter passes; a kernel that is not in it reports traffic: null
PMC_SQ_FILE = "r05_pmc_sq_summary.json"     # ... SQ counters per kernel (mfma_util, gpu_cycles, ...)
KERNEL_STATS_FILE = "r05_v1_kernel_stats.csv"   # ... rocprofv3 --kernel-trace --stats summary of `bench.py --pipeline 1`
# the same three files of the f16x3 plan (`--dtype f16x3`; tools/profile_round5.sh, profile_x3_pmc.sh, profile_x3_traffic.sh)
PROFILE_FILES_X3 = ("r05_f16x3_pmc_traffic.json", "r05_f16x3_pmc_sq_summary.json", "r05_f16x3_kernel_stats.csv")


class ClockSampler:
    """Shader clock during the timed region, as the kernel driver reports it (sysfs pp_dpm_sclk of this process's GPU: the entry
    marked `*`), sampled from a host thread every few milliseconds.  Best effort: a box that hides the file gives null.  The
    2.5 PFLOP/s peak assumes 2.4 GHz; under MFMA + LDS load on real data the boxes of this pool hold 1.9-2.2 GHz, and a reader of
    the roofline fractions needs to know which."""

    def __init__(self, dev_index):
        import glob
        import threading
        self.path = None
        try:
            pci = torch.cuda.get_device_properties(dev_index)
            want = getattr(pci, "pci_bus_id", None)
        except Exception:
            want = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        for c in cands:
            try:
                if want is not None:
                    bus = os.path.basename(os.path.realpath(os.path.dirname(c)))         # e.g. 0000:05:00.0
                    if int(bus.split(":")[1], 16) != int(want):
                        continue
                open(c).read()
                self.path = c
                break
            except Exception:
                continue
        if self.path is None and len(cands) == 1:
            self.path = cands[0]
        self.samples = []
        self._stop = threading.Event()
        self._thr = None

    def _read(self):
        try:
            for ln in open(self.path).read().splitlines():
                if "*" in ln:
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None

    def __enter__(self):
        import threading
        if self.path is not None:
            def loop():
                while not self._stop.is_set():
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.004)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr is not None:
            self._thr.join()

    def summary(self):
        if not self.samples:
            return None
        a = np.asarray(self.samples)
        return {"min": int(a.min()), "median": int(np.median(a)), "max": int(a.max()), "samples": int(a.size), "source": self.path}


def profile_figures(name):
    """What the committed profile of THIS round says about kernel `name` (rocprofv3 --kernel-trace --stats average duration; SQ
    counters from the separate --pmc passes): -> dict (missing files / kernels give nulls)."""
    out = {"avg_launch_ms_rocprof": None, "mfma_util": None, "clock_mhz_under_rocprof": None}
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", KERNEL_STATS_FILE)) as f:
            for row in csv.DictReader(f):
                import re
                nm = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", row.get("Name", ""))).strip()
                nm = re.sub(r"(, false)+>$", ">", nm)        # trailing defaulted template arguments (tools/pmc_summary.py norm)
                if nm == name:
                    out["avg_launch_ms_rocprof"] = round(float(row["AverageNs"]) / 1e6, 4)
                    break
    except Exception:
        pass
    try:
        k = json.load(open(os.path.join(ROOT, "profiles", PMC_SQ_FILE)))["kernels"].get(name)
        if k:
            out["mfma_util"] = k.get("mfma_util")
            if out["avg_launch_ms_rocprof"] and k.get("gpu_cycles"):
                out["clock_mhz_under_rocprof"] = int(round(k["gpu_cycles"] / (out["avg_launch_ms_rocprof"] * 1e3)))
    except Exception:
        pass
    return out


def op_flops(op):
    """Algorithmic FLOPs of one plan op (2 per MAC; conv/deconv only, SURVEY 8d convention)."""
    px = op.B * op.Ho * op.Wo
    if op.kind in (_lib.OP_CONV, _lib.OP_CONV_STREAM, _lib.OP_DCN, _lib.OP_STEM):
        return 2.0 * px * op.Cout * op.Cin * op.ksize * op.ksize
    if op.kind in (_lib.OP_DCN_FUSED, _lib.OP_DCN_FUSED_F16, _lib.OP_DCN_FUSED_STREAM):
        return 2.0 * px * (op.Cout + 27) * op.Cin * 9
    if op.kind == _lib.OP_UPDCN_F16:    # DeformConv (+ offset conv) at the output resolution, plus the 2x2-tap up-sampling
        return 2.0 * px * ((op.Cout + 27) * op.Cin * 9 + op.Cin * 4)
    if op.kind == _lib.OP_STEM3:        # 7x7 3->16 and 3x3 16->16 at full resolution, 3x3 16->32 at half (+ level2's 1x1 project 32->64 at a quarter)
        return 2.0 * op.B * (op.H * op.W * 16 * (3 * 49 + 16 * 9) + op.Ho * op.Wo * 32 * 16 * 9 + (op.Ho // 2) * (op.Wo // 2) * 64 * 32 * (1 if op.in2 else 0))
    if op.kind == _lib.OP_UPADD:
        return 2.0 * px * op.Cout * 4
    if op.kind == _lib.OP_HEADS:
        d = ctypes.cast(op.in2, ctypes.POINTER(_lib.H3dHeadsDesc)).contents
        return sum(2.0 * px * op.Cout * (op.Cin * 9 + d.head[i].C) for i in range(d.nheads))
    return 0.0


def op_bytes(op):
    """Algorithmic HBM bytes of one plan op: every operand read once, every result written once (weights excluded: they
    stay in L2 / the Infinity Cache across the batch)."""
    es = 4 if op.dtype in (_lib.H3D_F32, _lib.H3D_F16X3) else 2
    pin, pout = op.B * op.H * op.W, op.B * op.Ho * op.Wo
    if op.kind in (_lib.OP_STEM, _lib.OP_STEM3, _lib.OP_IM2COL):
        return 4.0 * pin * op.Cin + es * pout * op.Cout + (es * (pout // 4) * 64 if op.kind == _lib.OP_STEM3 and op.in2 else 0)
    if op.kind == _lib.OP_HEADS:
        d = ctypes.cast(op.in2, ctypes.POINTER(_lib.H3dHeadsDesc)).contents
        return es * pin * op.Cin + 4.0 * pout * sum(d.head[i].C for i in range(d.nheads))
    if op.kind == _lib.OP_UPADD:
        return es * (pin * op.Cin + 2.0 * pout * op.Cout)
    if op.kind == _lib.OP_UPDCN_F16:
        return es * (pin * op.Cin + pout * op.Cin + pout * op.Cout)
    out_es = 4 if op.out_mode in (_lib.OUT_NCHW_F32, _lib.OUT_NHWC_F32) else es
    b = es * pin * op.Cin + out_es * pout * op.Cout
    if op.kind in (_lib.OP_CONV, _lib.OP_CONV_STREAM) and op.in2:
        b += es * pout * op.Cout                   # residual
    return float(b)


def kernel_name(op):
    buf = ctypes.create_string_buffer(200)
    _lib.check(_lib.lib().h3d_op_kernel_name(ctypes.byref(op), buf, 200), "op_kernel_name")
    return buf.value.decode()


def per_kernel_profile(plan, iters):
    """HIP-event time of every op of the plan, grouped by kernel instantiation."""
    n = len(plan.ops)
    ms = (ctypes.c_float * n)()
    passes = []
    for _ in range(iters):
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "run_ops_timed")
        passes.append(np.frombuffer(ms, dtype=np.float32, count=n).copy())
    st = np.stack(passes)
    tot = np.median(st, axis=0)      # per launch: the median pass; the min / max passes are reported beside it
    lo, hi = st.min(axis=0), st.max(axis=0)
    groups = {}
    for i, op in enumerate(plan.ops):
        g = groups.setdefault(kernel_name(op), {"ms": 0.0, "ms_min": 0.0, "ms_max": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        g["ms"] += float(tot[i])
        g["ms_min"] += float(lo[i])
        g["ms_max"] += float(hi[i])
        g["flops"] += op_flops(op)
        g["bytes"] += op_bytes(op)
        g["launches"] += 1
    return groups


def usable_cores():
    """Host cores this process can really run on: the scheduler affinity, cut by the cgroup's CPU quota (a GPU box hands one GPU's
    job a 16-core share of a machine with far more cores: 200 threads on a 16-core quota are throttled to a crawl -- round 4's
    first cpu_baseline on "all cores" wrote nothing for 7 minutes), and by 16 when neither says less."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    else:
        n = min(n, 16)
    return max(1, n)


def cpu_baseline(opt, sd, keep=None):
    """Oracle (torch-CPU restatement, kind 'port') on a bounded sample of the same workload, as SURVEY 8(d) specifies it: fp32,
    batch 8, `torch.set_num_threads(N)` with N = ALL host cores this process may run on (`usable_cores`: affinity cut by the
    cgroup CPU quota; stated as `cores`, `host_cores` = what the machine has), warm-up 1, >= 5 timed iterations (about 10-15 s); then the 8-thread figure for comparison with the
    survey container's probe (4.2 images/s at B = 8, plain-conv variant), 3 iterations.
    keep: dict that receives the first 2 images of the sample and the oracle's head maps for them (index_match)."""
    from oracle import decode as odec, dla as odla, smpl as osmpl
    from h3d_amd import smpl as psmpl
    cores = usable_cores()
    net = odla.DLAOracle(sd, opt.heads, use_dcn=not opt.not_use_dcn)
    model = psmpl.SMPLModel.synthetic().numpy_dict()
    B = 8
    x = torch.from_numpy(synth.synth_images(B, opt.input_h, opt.input_w))

    def step():
        with torch.no_grad():
            o = {k: v.numpy() for k, v in net(x)[0].items()}
        if keep is not None and "heads" not in keep:
            keep["images"], keep["heads"] = x[:2].clone(), {k: v[:2].copy() for k, v in o.items()}
        dets, aux = odec.multi_pose_decode(odec.sigmoid_clamp(o["hm"]), o["wh"], o["hps"], o["reg"],
                                           odec.sigmoid_clamp(o["hm_hp"]), o["hp_offset"], K=opt.K, return_aux=True)
        n = 4                                             # meshes per image in the CPU sample (fp64 numpy)
        idx = aux["inds"][:, :n]
        th = np.stack([o["pose"][i].reshape(72, -1)[:, idx[i]].T for i in range(B)]).reshape(-1, 72)
        be = np.stack([o["shape"][i].reshape(10, -1)[:, idx[i]].T for i in range(B)]).reshape(-1, 10)
        osmpl.lbs(be, th, model)

    def run(threads, min_iters, budget):
        torch.set_num_threads(threads)
        t0 = time.time()
        step()                                            # warm-up
        print("[bench] cpu_baseline warm-up step %.1f s on %d threads" % (time.time() - t0, threads), file=sys.stderr, flush=True)
        t0 = time.time()
        iters = 0
        while iters < min_iters or (time.time() - t0 < budget and iters < 12):
            step()
            iters += 1
            print("[bench] cpu_baseline (%d threads) step %d: %.1f s elapsed" % (threads, iters, time.time() - t0), file=sys.stderr, flush=True)
        return B * iters / (time.time() - t0), iters

    v_all, it_all = run(cores, 5, 12.0)
    v8, it8 = (v_all, it_all) if cores == 8 else run(min(8, cores), 3, 6.0)
    torch.set_num_threads(cores)
    return {"value": round(v_all, 3), "unit": "images/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "value_8_threads": round(v8, 3),
            "sample": "%d steps of batch %d on %d threads (all cores this process may use), then %d steps on %d threads: 512x512, "
                      "DLA-34+DCNv2+heads fp32 via torch-CPU, numpy decode, 4 fp64 SMPL meshes/image"
                      % (it_all, B, cores, it8, min(8, cores))}


def frames_uint8(det, dev, batch, size, nslot, steps=20):
    """The loop the reference's path really starts with (datasets/coco_hp.py:151-212 -> H2D, trainer.py:259-261): `batch` raw
    uint8 BGR frames of size x size in PINNED host memory -> asynchronous H2D copy on the slot's stream -> `h3d_preprocess`
    (warp + normalise, csrc/preprocess.hip) -> network -> decode -> SMPL, `nslot` steps in flight (each slot its own pinned
    buffer, device frame buffer and plan buffers), so that the copy of step i+1 runs beside the kernels of step i.  Measured after
    the timed region; `value` stays the resident-in-HBM rate.  uint8 frames are 1/4 of the fp32 batch (50 MB instead of 201 MB
    per 64 images), which is what makes hiding the copy possible at all."""
    from h3d_amd import preprocess
    rng = np.random.default_rng(317)
    host = [torch.from_numpy(rng.integers(0, 256, size=(batch, size, size, 3), dtype=np.uint8)).pin_memory() for _ in range(nslot)]
    devb = [torch.empty(batch, size, size, 3, dtype=torch.uint8, device=dev) for _ in range(nslot)]
    streams = _streams(nslot, dev) if nslot > 1 else [torch.cuda.current_stream()]
    counter = [0]

    def step():
        k = counter[0] % nslot
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            devb[k].copy_(host[k], non_blocking=True)
            inp, c, s_ = preprocess.pre_process(devb[k], input_res=size)
            return det.run(inp, slot=k)["dets"]

    issued = []
    dt = time_steps(step, steps, nslot + 1, issued)
    # the two new stages alone, HIP events on one stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    h2d, pre = [], []
    for _ in range(5):
        ev[0].record()
        devb[0].copy_(host[0], non_blocking=True)
        ev[1].record()
        preprocess.pre_process(devb[0], input_res=size)
        ev[2].record()
        torch.cuda.synchronize()
        h2d.append(ev[0].elapsed_time(ev[1]))
        pre.append(ev[1].elapsed_time(ev[2]))
    h2d_ms, pre_ms = float(np.median(h2d)), float(np.median(pre))
    nbytes = batch * size * size * 3
    return {"images_per_s": round(batch * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "steps_in_flight": nslot,
            "batch": batch, "host_issue_ms_per_step": round(1e3 * issued[0] / steps, 3),
            "h2d_ms": round(h2d_ms, 3), "h2d_gb_per_s": round(nbytes / h2d_ms / 1e6, 1), "h2d_share_of_step": round(h2d_ms / (1e3 * dt / steps), 3),
            "preprocess_ms": round(pre_ms, 3),
            "note": "pinned uint8 frames -> H2D on the slot stream -> h3d_preprocess -> network -> decode -> SMPL; h2d_ms / preprocess_ms are the "
                    "stand-alone durations of those two stages (HIP events, one stream); with %d steps in flight the copy overlaps the "
                    "previous steps' kernels" % nslot}


def dcn_apron_stats(det, images, dev):
    """How far the DeformConv samples of THIS run's weights reach (synthetic offsets are small; trained networks have
    larger ones -- `--offset-scale`), counted by the kernels of the TIMED plan themselves on the TIMED batch: for every fused
    DeformConv op of `eng.plan(B, H, W)` -- with the (margin, patch slots) variant it really dispatches -- `h3d_dcn_far_samples`
    returns per 16x16 tile the number of (pixel, tap) samples that lie inside the image but have a bilinear corner outside
    the tile's LDS apron (such a sample takes one of the tile's NP patch slots, csrc/dcn3.hip).  `apron_miss_frac` = those
    samples / all samples; `tiles_over_slots_frac` = tiles with more of them than slots (only those run the slow
    global-gather pass 2).  (Round 3 evaluated an UNFUSED 2-image twin plan here, whose small grids pick other variants.)
    `mean_abs_offset_px` still comes from such a twin (bf16 plans only): the fused kernels never write their offsets."""
    import ctypes
    import re
    from h3d_amd.engine import Plan
    from h3d_amd._lib import H3dOp
    eng = det.model.engine(dev)
    B, _, H, W = images.shape
    plan = eng.plan(B, H, W)
    plan.op_array[0].in_ = images.data_ptr()
    plan.run()
    miss_all = tot_all = 0.0
    tiles_over = tiles_all = 0
    worst, worst_tiles, layers = 0.0, 0.0, {}
    for p, i in plan.dcn_layers:
        op = plan.ops[i]
        name = kernel_name(op)
        m = re.match(r"dcn3_kernel<[^,]+, \d+, \d+, (\d+), \d+, \w+, (\d+)", name)
        if m is None or op.Cin % 32 or op.reserved & 0x1000:
            return {"error": "no patch-slot DeformConv kernel behind %r (%s)" % (p, name)}
        margin, NP = int(m.group(1)), int(m.group(2))
        tiles = B * (-(-op.H // 16)) * (-(-op.W // 16))
        cnt = torch.empty(tiles, dtype=torch.int32, device=dev)
        q = H3dOp()
        ctypes.memmove(ctypes.byref(q), ctypes.byref(plan.op_array[i]), ctypes.sizeof(H3dOp))
        _lib.check(_lib.lib().h3d_dcn_far_samples(ctypes.byref(q), cnt.data_ptr(), _lib.stream_ptr()), "h3d_dcn_far_samples")
        torch.cuda.synchronize()
        over = int((cnt > NP).sum())
        miss, tot = float(cnt.sum()), 9.0 * B * op.H * op.W
        layers[p] = {"margin": margin, "slots": NP, "tiles": tiles, "tiles_over_slots": over, "apron_miss_frac": round(miss / tot, 5)}
        tiles_over, tiles_all = tiles_over + over, tiles_all + tiles
        worst_tiles = max(worst_tiles, over / tiles)
        miss_all, tot_all = miss_all + miss, tot_all + tot
        worst = max(worst, miss / tot)
    out = {"layers": len(layers), "batch": B, "source": "h3d_dcn_far_samples on the timed plan's ops and batch",
           "apron_miss_frac": round(miss_all / tot_all, 5), "worst_layer_apron_miss_frac": round(worst, 5),
           "tiles_over_slots_frac": round(tiles_over / max(tiles_all, 1), 5), "worst_layer_tiles_over_slots_frac": round(worst_tiles, 5),
           "per_layer": layers}
    if eng.pw.dtype == "bf16":
        twin = Plan(eng.pw, 2, H, W, **dict(eng._flags(), fuse_offsets=False))
        two = images[:2].contiguous()
        twin.op_array[0].in_ = two.data_ptr()
        twin.run()
        torch.cuda.synchronize()
        mean_abs = []
        for op in twin.ops:
            if op.kind == _lib.OP_DCN:
                om = [t for t in twin.keep if torch.is_tensor(t) and t.data_ptr() == op.in2][0]      # [2,h,w,32] fp32
                mean_abs.append(float(om[..., :18].abs().mean()))
        out["mean_abs_offset_px"] = round(float(np.mean(mean_abs)), 3)
        out["mean_abs_offset_px_source"] = "unfused twin plan on 2 images"
        del twin
    return out


def boundary_op_times(batch, dev):
    """The literal drop-in operator (`_ext.dcn_v2_forward`, DCNv2/src/dcn_v2.h:9-23) on the 16 DeformConv shapes of the network
     at this batch, HIP events on the current stream around the call, in three forms of the SAME Python function:
      reference  NCHW fp32 operands in, NCHW fp32 out (input relayout + offset/mask pack inside a workspace; since round 3 the
                 packed filters are cached per parameter version, as any caller running a layer repeatedly gets them);
      nhwc       the input in torch.channels_last memory format: read in place, channels-last fp32 out;
      bf16       a bfloat16 channels-last input: the network's bf16 DeformConv path (fp16 filters and blend, f16 MFMA).
    Not part of the timed step: the engine path never materialises offsets or NCHW tensors."""
    from h3d_amd import dcn_v2
    res = {}
    g = torch.Generator(device="cpu").manual_seed(0)
    tot = {"reference": 0.0, "nhwc": 0.0, "bf16": 0.0, "reference_f32_mfma": 0.0}

    def f32_mfma(fn):                    # the fp32 tensors' arithmetic of rounds 1-4 (exact fmaf chains on the fp32 matrix instruction)
        def run():
            dcn_v2.OP_F32_MFMA = True
            try:
                return fn()
            finally:
                dcn_v2.OP_F32_MFMA = False
        return run

    def timed(fn):
        for _ in range(2):
            fn()
        best = None
        for _ in range(3):               # fastest of three single calls (an allocation inside one call put 17 ms on one shape once)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        return best

    for prefix, o, ins, ups in arch.ida_specs():
        lvl = {"dla_up.ida_0": 32, "dla_up.ida_1": 64, "dla_up.ida_2": 128, "ida_up": 128}[prefix]     # output side at 512 x 512
        for k, (ci, f) in enumerate(zip(ins, ups), start=1):
            for name, cin, hw in (("proj_%d" % k, ci, lvl // f), ("node_%d" % k, o, lvl)):
                key = "%dx%d@%d" % (cin, o, hw)
                if key in res:
                    res[key]["n"] += 1
                    continue
                x = torch.randn(batch, cin, hw, hw, generator=g).to(dev)
                w = (torch.randn(o, cin, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
                b = torch.zeros(o, device=dev)
                off = (torch.randn(batch, 18, hw, hw, generator=g) * 1.5).to(dev)
                m = torch.rand(batch, 9, hw, hw, generator=g).to(dev)
                xcl = x.contiguous(memory_format=torch.channels_last)
                xbf = xcl.to(torch.bfloat16)
                flop = 2.0 * batch * hw * hw * o * cin * 9
                ms = {"reference": timed(lambda: dcn_v2.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)),
                      "nhwc": timed(lambda: dcn_v2.dcn_v2_forward(xcl, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)),
                      "bf16": timed(lambda: dcn_v2.dcn_v2_forward(xbf, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)),
                      "reference_f32_mfma": timed(f32_mfma(lambda: dcn_v2.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)))}
                res[key] = {"n": 1, "ms": round(ms["reference"], 3), "tflops": round(flop / ms["reference"] / 1e9, 1), "ms_f32_mfma": round(ms["reference_f32_mfma"], 3),
                            "ms_nhwc": round(ms["nhwc"], 3), "ms_bf16": round(ms["bf16"], 3), "tflops_bf16": round(flop / ms["bf16"] / 1e9, 1)}
                del x, w, off, m, xcl, xbf
    for v in res.values():
        tot["reference"] += v["ms"] * v["n"]
        tot["nhwc"] += v["ms_nhwc"] * v["n"]
        tot["bf16"] += v["ms_bf16"] * v["n"]
        tot["reference_f32_mfma"] += v["ms_f32_mfma"] * v["n"]
    return {"dtype": "reference / nhwc: fp32 tensors, every product as three fp16 MFMAs on split operands (f16x3, round 5; fp64-oracle error 4e-7 "
                     "relative, tests/test_gpu_dcn.py); ms_f32_mfma: the same call on v_mfma_f32_32x32x2_f32 (H3D_DCN_F32_MFMA, the rounds 1-4 "
                     "arithmetic, 7e-7); bf16: fp16 blend + v_mfma_f32_32x32x16_f16",
            "batch": batch, "layers": sum(v["n"] for v in res.values()),
            "total_ms": round(tot["reference"], 3), "total_ms_nhwc": round(tot["nhwc"], 3), "total_ms_bf16": round(tot["bf16"], 3),
            "total_ms_f32_mfma": round(tot["reference_f32_mfma"], 3),
            "shapes": res}


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD job (one process per GPU,
    rendezvous on 127.0.0.1) and relay its output; this process never touches the GPU, and nothing is exec'ed over a
    process that has.  Returns the child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """--dry-run: the launcher / rendezvous / barrier / max-over-ranks / JSON plumbing of the N-rank bench over gloo on
    CPU tensors -- no HIP call, no network step (tests/test_cpu_host.py).  The line is marked "dry_run": true."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    strong = args.global_batch > 0
    if strong:
        lo, hi = shard_batch(args.global_batch, rank, world)
        n_global = args.global_batch
    else:
        lo, hi, n_global = rank * args.batch, (rank + 1) * args.batch, world * args.batch
    dets = torch.full((hi - lo, 100, 40), float(rank))

    def step():
        return gather_detections(dets, n_images=n_global) if world > 1 else dets

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == (n_global, 100, 40)
    assert all(float(out[shard_batch(n_global, r, world)[0], 0, 0]) == r for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(n_global * args.steps / dt, 2), "unit": "images/s",
                          "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
                          "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "DRY RUN: all-gather of dets only (gloo, CPU)", "batch_per_gpu": hi - lo,
                                     "global_batch": n_global}}))
    if world > 1:
        dist.destroy_process_group()


def build_detector(arch_name, dtype, size, args, dev, people=None):
    """-> (detector, opt, synthetic state_dict, conv GFLOP per image) for one of the three backbones."""
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # (the factory prints "==> Use DeformConv." like the reference's, model.py:513;
        return _build_detector(arch_name, dtype, size, args, dev, people)      # stdout carries the ONE JSON line only)


def _build_detector(arch_name, dtype, size, args, dev, people=None):
    if arch_name == "dla_34":
        opt = Opt(input_h=size, input_w=size, smpl=True, smpl_people=args.people if people is None else people, dtype=dtype, K=100)
        sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, offset_scale=args.offset_scale, gain=args.weight_gain)
        det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
        return det, opt, sd, arch.conv_flops(opt.heads, True, size, size) / 1e9
    from h3d_amd import arch_hg, arch_res
    from h3d_amd.detector import make_detector
    if arch_name == "hourglass":
        opt = Opt(arch="hourglass", input_h=size, input_w=size, dtype=dtype, K=100)
        shapes, gain = arch_hg.state_dict_shapes(opt.heads), 0.8
        gflop_img = arch_hg.conv_flops(opt.heads, size, size) / 1e9
    else:
        opt = Opt(arch="resdcn_101", task="ctdet", input_h=size, input_w=size, dtype=dtype, K=100)
        shapes, gain = arch_res.state_dict_shapes(opt.heads, 64), 0.9
        gflop_img = arch_res.conv_flops(opt.heads, size, size) / 1e9
    sd = synth.synth_state_dict(shapes, seed=0, offset_scale=args.offset_scale, gain=gain)
    det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
    return det, opt, sd, gflop_img


def workload_name(arch_name, size, batch, strong, n_global):
    split = ("one batch of %d split over the ranks (this rank: %d)" % (n_global, batch)) if strong else ("batch %d per GPU" % batch)
    if arch_name == "dla_34":
        return ("DLA-34+DCNv2 multi_pose + pose/shape heads -> sigmoid/NMS/top-100 decode -> SMPL 6890-vert LBS; "
                "%dx%d, %s (BASELINE configs[2])" % (size, size, split))
    if arch_name == "hourglass":
        return "Hourglass-104 multi_pose -> decode; %dx%d, %s (BASELINE configs[3]: 128 over 8 GPUs = 16)" % (size, size, split)
    return "ResNet-101-DCN ctdet -> decode; %dx%d, %s (BASELINE configs[4]: 256 over 8 GPUs = 32)" % (size, size, split)


_STREAM_POOL = []


def _streams(n, dev):
    """The first n streams of one pool per process (stream -> hardware queue is decided at creation: re-creating streams for
    every measurement would change which of them share a queue)."""
    while len(_STREAM_POOL) < n:
        _STREAM_POOL.append(torch.cuda.Stream(device=dev))
    return _STREAM_POOL[:n]


def steps_in_flight_default(arch_name, batch):
    """3 (batch 64: 8420 / 8539 / 8537 images/s at 2 / 3 / 4 steps in flight, same box); small per-GPU shards (<= 16 images: grids
    of a few hundred workgroups) keep 4 so that the launches of different steps fill the CUs (batch 8: 3792 / 5525 / 6577 / 7090
    images/s at 1 / 2 / 3 / 4)."""
    if arch_name != "dla_34":
        return 4                 # Hourglass-104 16 x 512^2: 1476 / 1516 / 1495 images/s at 3 / 4 / 6; ResNet-101-DCN 32 x 768^2: 2672 / 2750 / 2772 at 2 / 3 / 4
    return 4 if batch <= 16 else 3


def make_step(det, images, nslot, world, n_global, dev, graph=False, rehearse=False):
    """-> (step(), per-image dets shape).  step() issues one batch: consecutive calls alternate between `nslot` HIP streams, each
    with its own copy of the plan's buffers; for world > 1 the single collective (all-gather of dets) follows on the
    default stream, in step order."""
    slot_streams = _streams(nslot, dev) if nslot > 1 else [None]
    counter = [0]
    dets_shape = (det.opt.K, 6 if det.opt.task == "ctdet" else 40)
    collective = world > 1 or rehearse          # rehearse: the N > 1 code path with a group of one rank (RCCL on a one-GPU box)

    def step():
        k = counter[0] % nslot
        counter[0] += 1
        kw = {"graph": True} if graph else {}
        if slot_streams[k] is None:
            res = det.run(images, slot=0, **kw)
            return gather_detections(res["dets"], n_images=n_global, force=rehearse) if collective else res["dets"]
        with torch.cuda.stream(slot_streams[k]):
            res = det.run(images, slot=k, **kw)
            if not collective:
                return res["dets"]
            done = torch.cuda.Event()
            done.record()
        # the collective is always issued from the default stream, in step order (one communicator, one stream)
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        res["dets"].record_stream(cur)
        return gather_detections(res["dets"], n_images=n_global, force=rehearse)

    return step, dets_shape


def time_steps(step, steps, warmup, issued=None):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if issued is not None:
        issued.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def shard_sweep(det, images, nslot, dev, batches, steps=200):
    """images/s of the SAME detector on the first B images of the batch, for the shard sizes a strong-scaling run of the
    headline batch would hand one GPU (64 over 8 / 4 / 2 GPUs).  Same step as the timed region (network + decode + SMPL,
    `nslot` steps in flight); measured after it, on one GPU."""
    out = {}
    for b in batches:
        if b >= images.shape[0]:
            continue
        sub = images[:b].contiguous()
        ns = steps_in_flight_default("dla_34", b)
        step, _ = make_step(det, sub, ns, 1, b, dev)
        issued = []
        st = steps if b <= 16 else max(30, steps * 16 // b)     # (a batch-8 step is ~1 ms: 30 steps were 33 ms, ramp and drain of the four
                                                                #  steps in flight included -- 5 % of the measurement)
        dt = time_steps(step, st, ns + 2, issued)             # (the first call of every slot lowers its plan: all inside the warm-up)
        out[str(b)] = {"images_per_s": round(b * st / dt, 1), "ms_per_step": round(1e3 * dt / st, 3),
                       "host_issue_ms_per_step": round(1e3 * issued[0] / st, 3), "steps_in_flight": ns, "steps": st}
    return out


def annotate_index_match(m, dtype):
    """`robust_prefix_equal` is vacuous when no rank is provably stable under the measured score error: say so."""
    m = dict(m)
    m["dtype"] = dtype
    if m["robust_prefix"] == 0:
        m["bit_match_evidence"] = ("none: no rank's order is provably stable under this plan's score error (robust_prefix 0), so "
                                   "robust_prefix_equal holds vacuously -- read agreement / set_overlap")
    else:
        m["bit_match_evidence"] = ("the first %d ranks cannot change order under the measured score error and are %s on the GPU"
                                   % (m["robust_prefix"], "bit-identical" if m["robust_prefix_equal"] else "NOT identical"))
    return m


def parity_mode(args, size, images, keep, dev, dtype="f32", steps=3, nslot=1):
    """The metric's second clause on the SAME workload: a parity plan timed on the same 64 images, and its top-k peak indices on the
    cpu_baseline's 2 images against the fp32 oracle.  dtype "f32" = exact fmaf chains on v_mfma_f32_32x32x2_f32; "f16x3" (round 5) =
    the same fp32 storage and launches with every fp32 product as three fp16 MFMAs on split operands (csrc/common.h ET<x3_t>).
    `verts_vs` (f16x3): max |vertex difference| against the f32 plan's meshes on the oracle's two images."""
    from oracle import index_match as oim
    det32, opt32, _, gflop = build_detector("dla_34", dtype, size, args, dev)
    step, _ = make_step(det32, images, nslot, 1, images.shape[0], dev)
    dt = time_steps(step, steps, nslot)
    res = det32.run(keep["images"].to(dev))
    m = annotate_index_match(oim.index_match({k: v.cpu().numpy() for k, v in res["heads"].items()}, res["inds"].cpu().numpy(),
                                             keep["heads"], K=opt32.K), dtype)
    out = {"dtype": dtype, "images_per_s": round(images.shape[0] * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "steps_in_flight": nslot, "batch": int(images.shape[0]),
           "step_frac_of_%s_peak" % ("f32_mfma" if dtype == "f32" else "f16x3"): round(gflop * images.shape[0] * steps / dt / 1e3 / PEAK_MFMA_TFLOPS[dtype], 4),
           "index_match": m}
    # the metric's third clause end to end: meshes against the CPU path's (the fp32 oracle's pose / shape maps read at the GPU's own
    # centre indices -> oracle/smpl.py in fp64); the north_star's tolerance is 1e-4 abs
    try:
        from oracle import smpl as osmpl
        n = res["verts"].shape[1]
        inds = res["inds"].cpu().numpy()[:, :n]
        th = np.concatenate([keep["heads"]["pose"][i].reshape(72, -1)[:, inds[i]].T for i in range(inds.shape[0])])
        be = np.concatenate([keep["heads"]["shape"][i].reshape(10, -1)[:, inds[i]].T for i in range(inds.shape[0])])
        v_ref, _ = osmpl.lbs(be, th, det32.smpl_model.numpy_dict())
        out["verts_max_abs_err_vs_oracle"] = float(np.abs(res["verts"].cpu().numpy().reshape(-1, v_ref.shape[1], 3) - v_ref).max())
    except Exception as e:
        out["verts_max_abs_err_vs_oracle"] = "%s: %s" % (type(e).__name__, e)
    if dtype == "f32":
        keep["verts_f32"] = res["verts"].clone()
        keep["inds_f32"] = res["inds"].clone()
    elif "verts_f32" in keep:
        out["inds_equal_f32_plan"] = bool(torch.equal(res["inds"], keep["inds_f32"]))
        out["verts_max_abs_diff_vs_f32_plan"] = float((res["verts"] - keep["verts_f32"]).abs().max()) if out["inds_equal_f32_plan"] else None
    del det32
    torch.cuda.empty_cache()
    return out


def fp16_plan(args, size, images, keep, dev, nslot, steps=10):
    """The fp16 plan (H3D_F16: fp16 activations and filters, saturating epilogues) on the SAME workload and images, and its top-k
    indices on the cpu_baseline's 2 images: three more significand bits than bf16 at the same MFMA rate."""
    from oracle import index_match as oim
    det16, opt16, _, gflop = build_detector("dla_34", "f16", size, args, dev)
    if args.dcn_margin == "auto":
        det16.model.engine(dev).calibrate_dcn_margins(images)
    step, _ = make_step(det16, images, nslot, 1, images.shape[0], dev)
    dt = time_steps(step, steps, nslot + 1)
    res = det16.run(keep["images"].to(dev))
    m = annotate_index_match(oim.index_match({k: v.cpu().numpy() for k, v in res["heads"].items()}, res["inds"].cpu().numpy(),
                                             keep["heads"], K=opt16.K), "f16")
    out = {"dtype": "f16", "images_per_s": round(images.shape[0] * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "steps_in_flight": nslot, "batch": int(images.shape[0]), "index_match": m}
    del det16
    torch.cuda.empty_cache()
    return out


def other_archs(args, dev, steps=5):
    """The per-GPU shards of BASELINE configs[3] / [4] (what `--arch hourglass` / `--arch resdcn_101` time), a few steps each
    after the headline measurement, so that the driver's one command observes them too: network + decode, no SMPL stage."""
    out = {}
    for name, batch, size, dtype, nslot in (("hourglass", 16, 512, "bf16", 4), ("resdcn_101", 32, 768, "f16", 4)):
        try:
            det, opt, _, gflop = build_detector(name, dtype, size, args, dev)
            images = torch.from_numpy(synth.synth_image_batch(batch, size, size, seed=317)).to(dev)
            step, _ = make_step(det, images, nslot, 1, batch, dev)
            dt = time_steps(step, steps, nslot + 1)          # (every slot's plan is lowered inside the warm-up)
            out[name] = {"images_per_s": round(batch * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "batch_per_gpu": batch,
                         "size": size, "dtype": dtype, "steps": steps, "steps_in_flight": nslot,
                         "model_tflops": round(gflop * batch * steps / dt / 1e3, 1)}
            del det, images
            torch.cuda.empty_cache()
        except Exception as e:          # a record measured AFTER the headline number must not take the line down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def offset_robustness(args, size, batch, dev, nslot, value, steps=10):
    """The headline workload with LARGER DeformConv offsets than the synthetic default (`--offset-scale` 0.5: mean |offset| 2.0 px):
    scale 1.0 (3.8 px: a quarter of the tiles have more far samples than the default variant's 256 slots) and 2.0 (6.5 px: most
    do).  Trained networks' offsets are routinely larger than 2 px, so the headline alone would be an optimistic point.  Same
    step, same batch, the per-layer variant rule applied to the new weights; measured after the timed region."""
    import copy
    out = {}
    for sc in (1.0, 2.0):
        try:
            a2 = copy.copy(args)
            a2.offset_scale = sc
            det, opt, _, _ = build_detector("dla_34", args.dtype, size, a2, dev)
            images = torch.from_numpy(synth.synth_image_batch(batch, size, size, seed=317)).to(dev)
            eng = det.model.engine(dev)
            rep = eng.calibrate_dcn_margins(images) if args.dcn_margin == "auto" else {}
            step, _ = make_step(det, images, nslot, 1, batch, dev)
            dt = time_steps(step, steps, nslot + 1)
            names = {v: k for k, v in eng.DCN_VARIANTS.items()}
            choice = [names[v] for v in eng.pw.dcn_variant.values()]
            out["%.1f" % sc] = {"images_per_s": round(batch * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3),
                                "frac_of_headline": round(batch * steps / dt / value, 3), "steps": steps,
                                "layers_on_slots512": choice.count("slots512"), "layers_on_wide": choice.count("wide"),
                                "tiles_over_256_mean": round(float(np.mean([r["tiles_over_256"] for r in rep.values()])), 3) if rep else None}
            del det, eng, images
            torch.cuda.empty_cache()
        except Exception as e:          # a record measured AFTER the headline number must not take the line down with it
            out["%.1f" % sc] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


METRIC = "images/sec whole-node, DLA-34+SMPL batch-64 512x512; top-k index bit-match"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (weak scaling: the default)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="> 0: STRONG scaling -- one batch of this many images per step for the whole job, rank r takes the "
                         "contiguous slice detector.shard_batch(G, r, world) (the reference's DataParallel scatter of ONE batch, "
                         "trains/trainer.py:176; SURVEY 8e: batch 64 -> 8 per GPU on 8 GPUs); the line then says \"scaling\": \"strong\"")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay every step as one hipGraph (detector.run(graph=True)); 0 (default): issue every launch from Python. "
                         "Measured: no gain -- batch 8, 1 / 2 / 3 steps in flight: 3792 / 5526 / 6373 images/s with graphs, 3760 / 5512 / "
                         "6320 without; batch 64: 8180 vs 8320 (the host issues a step faster than the GPU runs it, even at 8 images)")
    ap.add_argument("--dcn-margin", default="auto", choices=["narrow", "slots512", "wide", "auto"],
                    help="DeformConv tile variant: narrow = margin 2, 256 patch slots (fastest while offsets are small), slots512 = margin 2 "
                         "with 512 slots in two rounds, wide = margin 4 on the packed apron, each for every layer; auto (default; bf16 / f16 "
                         "DLA-34 plans) = per layer, by a deterministic rule on the kernels' own per-tile far-sample counts of the job's batch, outside the timed region "
                         "(DLAEngine.calibrate_dcn_margins).  Same box, images/s at --offset-scale 0.5 / 1.0 / 2.0: narrow "
                         "8310 / 6831 / 5768, wide 7880 / 7143 / 6095, auto 8301 / 7340 / 6082")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="one GPU only: run the step exactly as a multi-GPU rank does -- RCCL process group (of one rank), the per-step "
                         "all_gather_into_tensor of dets issued from the default stream behind an event of the slot stream, barrier + "
                         "max-over-ranks timing -- so that the N > 1 code path meets the device before the driver's 8-GPU run does")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the records measured after the timed region (shard_sweep, parity_mode, other_archs)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f16x3"])
    ap.add_argument("--people", type=int, default=100, help="SMPL meshes per image (<= K)")
    ap.add_argument("--streams", type=int, default=1, help="sub-batches run concurrently on their own HIP streams")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="steps in flight (default 3; 4 for per-GPU batches of at most 16 images): consecutive batches alternate between this many HIP streams, each with its own copy "
                         "of the plan's buffers, so the decode / SMPL tail of one batch (small grids) overlaps the network of "
                         "the next.  Same launches, same work per step; 1 = strictly one batch at a time")
    ap.add_argument("--offset-scale", type=float, default=0.5,
                    help="scale of the synthetic conv_offset_mask filters (h3d_amd.synth): DCN offsets are ~N(0, (0.6 S)^2) px; "
                         "trained networks have larger offsets than the default, which moves samples into the DeformConv's "
                         "global-gather pass 2")
    ap.add_argument("--weight-gain", type=float, default=1.25,
                    help="gain of the synthetic conv weights (h3d_amd.synth): 1.25 keeps the signal alive through DLA-34, so "
                         "the heat map has separated peaks and index_match means something; 1.0 = round 1's weights")
    ap.add_argument("--arch", default="dla_34", choices=["dla_34", "hourglass", "resdcn_101"],
                    help="dla_34 = the headline workload (BASELINE configs[1]/[2]); hourglass = configs[3] (multi_pose, 512x512, "
                         "16 images per GPU); resdcn_101 = configs[4] (ctdet, 768x768, 32 images per GPU): per-GPU shard of the "
                         "8-GPU batch, network + decode, no SMPL stage, no CPU baseline")
    ap.add_argument("--size", type=int, default=0, help="input height = width (default 512; 768 for resdcn_101)")
    ap.add_argument("--engine-flag", action="append", default=[], help="engine lowering flag NAME=INT (engine.Plan.FLAGS), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="launcher + collective plumbing only, gloo on CPU (tests)")
    args = ap.parse_args()
    if args.dtype == "f16x3":          # the roofline record reads the f16x3 plan's own committed profile
        global PMC_TRAFFIC_FILE, PMC_SQ_FILE, KERNEL_STATS_FILE
        PMC_TRAFFIC_FILE, PMC_SQ_FILE, KERNEL_STATS_FILE = PROFILE_FILES_X3

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.rehearse_collective:
        import torch.distributed as dist
        if world == 1:                                      # a group of one rank, started without torchrun
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm

    dla = args.arch == "dla_34"
    size = args.size or (768 if args.arch == "resdcn_101" else 512)
    strong = args.global_batch > 0
    if strong:
        lo, hi = shard_batch(args.global_batch, rank, world)
        batch, n_global = hi - lo, args.global_batch
        if batch == 0:
            raise SystemExit("--global-batch %d leaves rank %d of %d without images" % (args.global_batch, rank, world))
    else:
        lo, batch, n_global = rank * args.batch, args.batch, world * args.batch
    det, opt, sd, gflop_img = build_detector(args.arch, args.dtype, size, args, dev)
    eng = det.model.engine(dev)
    eng.streams = args.streams
    for kv in args.engine_flag:
        name, val = kv.split("=")
        setattr(eng, name, int(val))
    # every image of the job is a different synthetic image (image i of the global batch is a pure function of i)
    images = torch.from_numpy(synth.synth_image_batch(batch, size, size, seed=317, first=lo)).to(dev)
    dcn_variants = None
    if args.dcn_margin == "wide":
        eng.dcn_wide_margin = 1
    elif args.dcn_margin == "slots512":
        eng.dcn_slots512 = 1
    elif args.dcn_margin == "auto" and args.dtype in ("bf16", "f16") and dla:
        # setup, outside the timed region: the per-layer tile variant from the kernels' own far-sample counts on this batch -- a
        # deterministic rule (engine.DLAEngine.calibrate_dcn_margins), not a stopwatch: a second process makes the same choice
        eng.calibrate_dcn_margins(images)
        names = {v: k for k, v in eng.DCN_VARIANTS.items()}
        dcn_variants = {p: names[v] for p, v in sorted(eng.pw.dcn_variant.items())}

    nslot = max(1, args.pipeline if args.pipeline is not None else steps_in_flight_default(args.arch, batch))
    if args.streams > 1 and args.pipeline is None:
        nslot = 1              # sub-batch streams and plan slots are two uses of the same idea: combined only on request
    use_graph = dla and args.graph == 1 and args.streams <= 1
    step, dets_shape = make_step(det, images, nslot, world, n_global, dev, graph=use_graph, rehearse=args.rehearse_collective and world == 1)

    # one-time setup outside both warm-up and the timed region: weight packing + plan lowering (host work and uploads,
    # no network launches) and the SMPL model upload.  The first launch of every kernel still pays its code-object
    # load, so keep --warmup >= 1
    t_setup = time.perf_counter()
    if args.streams <= 1:                       # (the sub-batch plans of --streams N are built by the first step)
        for k in range(nslot):
            eng.plan(batch, size, size, k)
    from h3d_amd import smpl as _smpl
    if dla and det.smpl_model._dev is None:
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
    clock = ClockSampler(local)
    with clock:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        t_issued = time.perf_counter() - t0        # host time to ISSUE the K steps (launch-bound when it approaches dt)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == (n_global,) + dets_shape, (out.shape, n_global, dets_shape)
    if rank == 0:
        print("[bench] %d GPU(s): %.1f images/s, %.3f ms/step" % (world, n_global * args.steps / dt,
                                                                 1e3 * dt / args.steps), file=sys.stderr, flush=True)
    # a second, longer measurement of the same step (the driver fixes --steps: 20 steps are 0.15 s, boxes of the pool differ by a few
    # percent over such a window).  Never `value`.
    long_steps = max(200, args.steps) if (dla and not args.no_extras) else 0
    value_long = None
    if long_steps:
        clock_long = ClockSampler(local)
        with clock_long:
            if dist is not None:
                dist.barrier()
            dt_long = time_steps(step, long_steps, 0)
            if dist is not None:
                dist.barrier()
                t = torch.tensor([dt_long], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_long = float(t.item())
        value_long = {"images_per_s": round(n_global * long_steps / dt_long, 2), "steps": long_steps, "ms_per_step": round(1e3 * dt_long / long_steps, 3),
                      "shader_clock_mhz": clock_long.summary()}

    if rank == 0:
        peak = PEAK_MFMA_TFLOPS[args.dtype]
        line = {
            "metric": METRIC,
            "value": round(n_global * args.steps / dt, 2), "unit": "images/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "host_issue_ms_per_step": round(1e3 * t_issued / args.steps, 3),
            "value_long": value_long, "shader_clock_mhz": clock.summary(),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_name(args.arch, size, batch, strong, n_global),
                       "batch_per_gpu": batch, "global_batch": n_global, "K": 100,
                       "smpl_people_per_image": args.people if dla else 0, "conv_gflop_per_image": round(gflop_img, 2),
                       "parallelism": "dp%d (image shards, one all-gather of dets)" % world,
                       "collective": ("rehearsed: RCCL group of one rank, all_gather_into_tensor issued every step" if (args.rehearse_collective and world == 1)
                                      else "all_gather_into_tensor of dets, every step" if world > 1 else "none (one rank)"),
                       "steps_in_flight": nslot, "hip_graph": bool(use_graph), "dcn_margin": args.dcn_margin,
                       "dcn_variants": dcn_variants,
                       "images": "%d distinct synthetic images (h3d_amd.synth.synth_image_batch, seed 317)" % n_global,
                       "weights": "synthetic (h3d_amd.synth, seed 0, gain %s, offset_scale %g)"
                                  % ("%g" % args.weight_gain if dla else "per arch", args.offset_scale)},
            # what the metric's second clause ("top-k index bit-match") can mean per arithmetic: see index_match / parity_mode
            "metric_note": "value = throughput of the %s plan; top-k index bit-match against the fp32 oracle is attainable in the fp32-storage "
                           "plans (parity_mode: f32 = exact fmaf chains, f16x3 = the same on split-operand fp16 MFMAs at ~3x the rate), the %s plan is "
                           "reported with set overlap / agreement (index_match)" % (args.dtype, args.dtype),
        }
        line["model_tflops"] = round(gflop_img * line["value"] / 1e3 / world, 1)      # per GPU
        if not args.no_roofline:
            plan = eng.plan(batch, size, size)
            groups = per_kernel_profile(plan, iters=5)
            total_ms = sum(g["ms"] for g in groups.values())
            # dominant kernel = the largest instantiation of the kernel FAMILY (template name) with the most device time: with all
            # heads in one launch that single launch outweighs any one DeformConv instantiation, but the DeformConv family as a
            # whole (three instantiations) is still where most of the step goes -- and it is the one furthest from its roofline
            fam_ms = {}
            for k, v in groups.items():
                fam_ms[k.split("<")[0]] = fam_ms.get(k.split("<")[0], 0.0) + v["ms"]
            family = max(fam_ms, key=fam_ms.get)
            fam_flops = sum(v["flops"] for k, v in groups.items() if k.split("<")[0] == family)
            name, g = max(((k, v) for k, v in groups.items() if k.split("<")[0] == family), key=lambda kv: kv[1]["ms"])
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            traffic, traffic_src = None, None   # HBM bytes per launch from THIS round's committed rocprofv3 --pmc passes, or null
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))["kernels"]
                traffic = pmc.get(name, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError, KeyError):
                traffic = None
            if traffic:
                traffic_src = ("profiles/%s (FETCH_SIZE + WRITE_SIZE, separate --pmc passes, gfx950 unit corrections)" % PMC_TRAFFIC_FILE)
            net_tflops = batch * gflop_img / total_ms                         # GFLOP / ms = TFLOP/s
            line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1),
                                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": g["launches"],
                                "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                                "avg_launch_ms_min_max": [round(g["ms_min"] / g["launches"], 4), round(g["ms_max"] / g["launches"], 4)],
                                "share_of_network_time": round(g["ms"] / total_ms, 3),
                                "family": family, "family_ms_per_step": round(fam_ms[family], 3),
                                "family_frac": round(fam_flops / (fam_ms[family] * 1e-3) / 1e12 / peak, 4),
                                "family_share_of_network_time": round(fam_ms[family] / total_ms, 3),
                                "network_ms_per_step": round(total_ms, 3),
                                # the north_star's target is quoted on the whole DLA-34+DCNv2 forward: all conv FLOP of
                                # the network / its device time, and the same FLOP / the whole step (decode, SMPL, gather)
                                "timing": "HIP events around every launch of the plan on one stream, median of 5 passes per launch, min / max pass beside it "
                                          "(independent of steps_in_flight: kernels of two steps sharing the GPU stretch individual launches)",
                                "network_tflops": round(net_tflops, 1), "network_frac": round(net_tflops / peak, 4),
                                "step_frac": round(line["model_tflops"] / peak, 4)}
            # what a reader needs to tell a box difference from a kernel change: the shader clock of the timed region, and this
            # round's committed profile of the same kernel (rocprofv3 average duration -> frac_rocprof; MFMA pipe utilisation)
            pf = profile_figures(name)
            line["roofline"].update(pf)
            line["roofline"]["frac_rocprof"] = (round(g["flops"] / g["launches"] / (pf["avg_launch_ms_rocprof"] * 1e-3) / 1e12 / peak, 4)
                                                if pf["avg_launch_ms_rocprof"] else None)
            line["roofline"]["shader_clock_mhz_timed_region"] = line["shader_clock_mhz"]
            line["roofline"]["peak_assumes_mhz"] = 2400
            print("[bench] roofline %s" % json.dumps(line["roofline"]), file=sys.stderr, flush=True)
            line["kernels"] = {k: {"ms": round(v["ms"], 3), "ms_min": round(v["ms_min"], 3), "ms_max": round(v["ms_max"], 3), "n": v["launches"],
                                   "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                                   "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 0)}      # algorithmic HBM bytes / time
                               for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}
            print("[bench] kernels %s" % json.dumps(line["kernels"]), file=sys.stderr, flush=True)
            if dla and args.dtype in ("bf16", "f16") and opt.not_use_dcn is False:
                line["dcn_apron"] = dcn_apron_stats(det, images, dev)
            if world == 1 and dla:
                line["boundary_op"] = boundary_op_times(min(batch, 16), dev)
                print("[bench] boundary_op %s" % json.dumps(line["boundary_op"]), file=sys.stderr, flush=True)
        extras = world == 1 and dla and not args.no_extras and not strong and args.dtype == "bf16"
        if extras:
            # the per-GPU shards a STRONG-scaling run of the headline batch would see (64 over 8 GPUs = 8 per GPU)
            line["shard_sweep"] = shard_sweep(det, images, nslot, dev, (8, 16, 32))
            print("[bench] shard_sweep %s" % json.dumps(line["shard_sweep"]), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1 and dla:
            keep = {}
            line["cpu_baseline"] = cpu_baseline(opt, sd, keep=keep)
            # the checker's second use: the same 2 images through the GPU path, indices compared with the oracle's
            from oracle import index_match as oim
            res = det.run(keep["images"].to(dev))
            line["index_match"] = annotate_index_match(oim.index_match({k: v.cpu().numpy() for k, v in res["heads"].items()},
                                                                       res["inds"].cpu().numpy(), keep["heads"], K=opt.K), args.dtype)
            print("[bench] index_match %s" % json.dumps(line["index_match"]), file=sys.stderr, flush=True)
            if extras:
                try:
                    line["frames_uint8"] = frames_uint8(det, dev, batch, size, nslot)
                except Exception as e:          # a record measured AFTER the headline number must not take the line down with it
                    line["frames_uint8"] = {"error": "%s: %s" % (type(e).__name__, e)}
                print("[bench] frames_uint8 %s" % json.dumps(line["frames_uint8"]), file=sys.stderr, flush=True)
                line["parity_mode"] = parity_mode(args, size, images, keep, dev)
                print("[bench] parity_mode %s" % json.dumps(line["parity_mode"]), file=sys.stderr, flush=True)
                try:        # the same contract on the fp16 matrix cores (three steps in flight like the headline; 20 steps)
                    line["parity_mode"]["f16x3"] = parity_mode(args, size, images, keep, dev, dtype="f16x3", steps=20, nslot=nslot)
                except Exception as e:          # a record measured AFTER the headline number must not take the line down with it
                    line["parity_mode"]["f16x3"] = {"error": "%s: %s" % (type(e).__name__, e)}
                keep.pop("verts_f32", None)
                print("[bench] parity_mode f16x3 %s" % json.dumps(line["parity_mode"]["f16x3"]), file=sys.stderr, flush=True)
                px = line["parity_mode"]["f16x3"]
                if "images_per_s" in px:
                    line["metric_note"] += (
                        "  PARITY PLAN f16x3 (Opt(dtype='f16x3'): fp32 storage, three fp16 MFMAs on split operands per fp32 product), same 64 images: "
                        "%.1f images/s (%.2fx the f32 plan's %.1f), max head error vs the fp32 oracle %.2g (f32 plan: %.2g), top-k agreement %.3f, "
                        "robust prefix %d bit-identical, meshes within %s of the CPU path's (tolerance 1e-4)."
                        % (px["images_per_s"], px["images_per_s"] / line["parity_mode"]["images_per_s"], line["parity_mode"]["images_per_s"],
                           px["index_match"]["max_abs_head_err"], line["parity_mode"]["index_match"]["max_abs_head_err"], px["index_match"]["agreement"],
                           px["index_match"]["robust_prefix"],
                           ("%.2g" % px["verts_max_abs_err_vs_oracle"]) if isinstance(px.get("verts_max_abs_err_vs_oracle"), float) else "n/a"))
                if args.dtype == "bf16":
                    try:
                        line["fp16_plan"] = fp16_plan(args, size, images, keep, dev, nslot)
                    except Exception as e:      # a record measured AFTER the headline number must not take the line down with it
                        line["fp16_plan"] = {"error": "%s: %s" % (type(e).__name__, e)}
                    print("[bench] fp16_plan %s" % json.dumps(line["fp16_plan"]), file=sys.stderr, flush=True)
                    f16 = line["fp16_plan"]
                    if "images_per_s" in f16:
                        # co-headline: the SAME workload in fp16 (same MFMA rate, three more significand bits per stored activation)
                        im16, im = f16["index_match"], line["index_match"]
                        line["metric_note"] += (
                            "  CO-HEADLINE fp16 plan (Opt(dtype='f16')): %.1f images/s (%.3f of the bf16 value), max head error vs the fp32 oracle "
                            "%.3g (bf16: %.3g), top-k set overlap %.3f (bf16: %.3f), positional agreement %.3f (bf16: %.3f).  Which to run: "
                            "bf16 is the arithmetic north_star prices the roofline in and stays `value`; run fp16 when the ranking of the "
                            "peaks matters more than ~2 %% of throughput (the stored activations of DLA-34 stay far inside +-65504 and the fp16 "
                            "epilogues saturate), and f16x3 (or f32: parity_mode) when indices must match the reference bit for bit."
                            % (f16["images_per_s"], f16["images_per_s"] / line["value"], im16["max_abs_head_err"], im["max_abs_head_err"],
                               im16["set_overlap"], im["set_overlap"], im16["agreement"], im["agreement"]))
        if extras:
            del det, eng, images
            torch.cuda.empty_cache()
            line["offset_robustness"] = offset_robustness(args, size, batch, dev, nslot, line["value"])
            print("[bench] offset_robustness %s" % json.dumps(line["offset_robustness"]), file=sys.stderr, flush=True)
            line["other_archs"] = other_archs(args, dev)
            print("[bench] other_archs %s" % json.dumps(line["other_archs"]), file=sys.stderr, flush=True)
        if "index_match" in line:      # LAST on stderr: a 2000-character tail of this log shows the HEADLINE plan's parity, not only the co-plans'
            print("[bench] headline plan (%s) index_match %s" % (args.dtype, json.dumps(line["index_match"])), file=sys.stderr, flush=True)
            print("[bench] headline %.1f images/s (%s), value_long %s, shader clock %s" % (line["value"], args.dtype, json.dumps(value_long), json.dumps(line["shader_clock_mhz"])),
                  file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
