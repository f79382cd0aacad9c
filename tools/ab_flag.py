"""Profiling aid: A/B the values of ONE engine lowering flag (engine.Plan.FLAGS) on the bench plan inside one process: a plan
per value, every launch timed with HIP events (`h3d_run_ops_timed`), values alternating.

    python tools/ab_flag.py mixed_heads 0 1 [--filter heads] [--batch 64] [--dtype bf16]
"""
import argparse, ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name

ap = argparse.ArgumentParser()
ap.add_argument("flag")
ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--filter", default="")
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype=args.dtype)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
eng = det.model.engine(dev)
x = torch.from_numpy(synth.synth_image_batch(args.batch, 512, 512)).to(dev)
plans = {}
for v in args.values:
    setattr(eng, args.flag, v)
    eng.plans.clear()
    det.run(x); torch.cuda.synchronize()
    plans[v] = eng.plan(args.batch, 512, 512)
runs = {v: [] for v in plans}
for rep in range(args.reps + 1):
    for v, plan in plans.items():
        n = len(plan.ops)
        ms = (ctypes.c_float * n)()
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "h3d_run_ops_timed")
        if rep:
            runs[v].append(np.frombuffer(ms, dtype=np.float32, count=n).copy())
for v, plan in plans.items():
    med = np.median(np.stack(runs[v]), axis=0)
    fam = {}
    for i, op in enumerate(plan.ops):
        nm = kernel_name(op)
        if args.filter in nm:
            fam.setdefault(nm, []).append(med[i])
    print("%s = %d: %d launches, %.4f ms in all; %s: %.4f ms" % (args.flag, v, len(plan.ops), med.sum(), args.filter or "all",
                                                            sum(sum(t) for t in fam.values())))
    for nm, t in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        print("    %-72s %3d  %.4f" % (nm[:72], len(t), sum(t)))
