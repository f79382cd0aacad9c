"""Profiling aid: time the ops of one kind of a plan under several `h3d_op.reserved` tuning overrides inside ONE process.

    python tools/ab_op_reserved.py --dtype f16x3 --kind 9 --codes 0 0x2000 0x4000        # H3D_OP_DCN_FUSED of the f16x3 plan: margin 4 / 2 / 6
"""
import argparse, ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd  # noqa: F401
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16x3")
ap.add_argument("--kind", type=int, default=9)
ap.add_argument("--codes", nargs="+", default=["0"])
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--offset-scale", type=float, default=0.5)
args = ap.parse_args()
codes = [int(c, 0) for c in args.codes]
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype=args.dtype)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25, offset_scale=args.offset_scale)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_image_batch(args.batch, 512, 512)).to(dev)
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(args.batch, 512, 512)
ops = [i for i, op in enumerate(plan.ops) if op.kind == args.kind]
shapes = sorted({(plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) for i in ops})
ms = (ctypes.c_float * 1)()
print("%-22s" % "Cin->Cout @H (n)" + "".join("%10s" % hex(c) for c in codes))
tot = [0.0] * len(codes)
for sh in shapes:
    idx = [i for i in ops if (plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) == sh]
    row = []
    for c in codes:
        t = 0.0
        for i in idx:
            op = _lib.H3dOp()
            ctypes.memmove(ctypes.byref(op), ctypes.byref(plan.op_array[i]), ctypes.sizeof(_lib.H3dOp))
            op.reserved = c
            arr = (_lib.H3dOp * 1)(op)
            ts = []
            for _ in range(args.reps + 1):
                _lib.check(_lib.lib().h3d_run_ops_timed(arr, 1, _lib.stream_ptr(), ms), "timed")
                ts.append(ms[0])
            t += float(np.median(ts[1:]))
        row.append(t)
    tot = [a + b for a, b in zip(tot, row)]
    print("%-22s" % ("%d->%d @%d (%d)" % (sh + (len(idx),))) + "".join("%10.4f" % v for v in row))
print("%-22s" % "total" + "".join("%10.4f" % v for v in tot))
