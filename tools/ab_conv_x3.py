"""Profiling aid: tile configurations of the f16x3 3x3 convolutions (csrc/conv.hip launch_conv_t<x3_t>), A/B'd inside one process on the
ops of the batch-64 f16x3 plan: reserved = 0x1000 | MT << 8 | WAVES << 4 | TH >> 3 (0 = the launcher's own choice).

    python tools/ab_conv_x3.py [--batch 64]
"""
import argparse, ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd  # noqa: F401
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--ksize", type=int, default=3)
ap.add_argument("--min-cout", type=int, default=64)
ap.add_argument("--codes", nargs="*", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="f16x3")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_image_batch(args.batch, 512, 512)).to(dev)
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(args.batch, 512, 512)
ops = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV and op.ksize == args.ksize and op.stride == args.stride and op.Cout >= args.min_cout]
shapes = sorted({(plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) for i in ops})
codes = [int(c, 0) for c in args.codes] if args.codes else [0, 0x1242, 0x1282, 0x1482, 0x1441, 0x1241]
ms = (ctypes.c_float * 1)()
print("%-22s" % "Cin->Cout @H (n)" + "".join("%10s" % ("auto" if c == 0 else hex(c)) for c in codes))
for sh in shapes:
    idx = [i for i in ops if (plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) == sh]
    row = []
    for c in codes:
        tot = 0.0
        ok = True
        for i in idx:
            op = _lib.H3dOp()
            ctypes.memmove(ctypes.byref(op), ctypes.byref(plan.op_array[i]), ctypes.sizeof(_lib.H3dOp))
            op.reserved = c
            arr = (_lib.H3dOp * 1)(op)
            ts = []
            for _ in range(args.reps + 1):
                rc = _lib.lib().h3d_run_ops_timed(arr, 1, _lib.stream_ptr(), ms)
                if rc:
                    ok = False
                    break
                ts.append(ms[0])
            if not ok:
                break
            tot += float(np.median(ts[1:]))
        row.append(tot if ok else float("nan"))
    print("%-22s" % ("%d->%d @%d (%d)" % (sh + (len(idx),))) + "".join("%10.4f" % v for v in row))
