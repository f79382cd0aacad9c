"""Profiling aid: per-kernel time of ONE step of the Hourglass-104 / ResNet-101-DCN shards of BASELINE configs[3] / [4] (launch by launch,
HIP events around each launch: h3d_run_ops_timed), with the algorithmic FLOP rate and bytes of each launch.

    python tools/arch_kernels.py resdcn_101 [--dtype f16] [--batch 32] [--size 768]
    python tools/arch_kernels.py hourglass [--dtype bf16] [--batch 16] [--size 512]
"""
import argparse, ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd  # noqa: F401
from h3d_amd import _lib, synth
from bench import build_detector, kernel_name, op_bytes, op_flops

ap = argparse.ArgumentParser()
ap.add_argument("arch")
ap.add_argument("--dtype", default=None)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--size", type=int, default=None)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--per-op", action="store_true")
ap.add_argument("--codes", nargs="*", default=None)
ap.add_argument("--offset-scale", type=float, default=1.0)
ap.add_argument("--weight-gain", type=float, default=1.25)
ap.add_argument("--people", type=int, default=100)
args = ap.parse_args()
hg = args.arch == "hourglass"
dtype = args.dtype or ("bf16" if hg else "f16")
batch = args.batch or (16 if hg else 32)
size = args.size or (512 if hg else 768)
dev = torch.device("cuda:0")
det, opt, sd, gflop = build_detector(args.arch, dtype, size, args, dev)
x = torch.from_numpy(synth.synth_image_batch(batch, size, size)).to(dev)
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(batch, size, size)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
runs = []
for _ in range(args.reps + 1):
    _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "run_ops_timed")
    runs.append(np.array(ms[:]))
t = np.median(np.stack(runs[1:]), axis=0)
rows = {}
for i, op in enumerate(plan.ops):
    key = kernel_name(op) + ((" %d->%d @%dx%d k%d s%d" % (op.Cin, op.Cout, op.H, op.W, op.ksize, op.stride)) if args.per_op else "")
    r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
    r[0] += 1; r[1] += float(t[i]); r[2] += op_flops(op); r[3] += op_bytes(op)
tot = float(t.sum())
print("%s %s batch %d %dx%d: %d launches, %.3f ms (%.1f images/s if nothing else ran), %.1f GFLOP/image" % (args.arch, dtype, batch, size, size, n, tot, batch / tot * 1e3, gflop))
for k, (c, m, f, b) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-86s %3d %8.3f ms %5.1f%% %7.1f TF/s %6.0f GB/s" % (k[:86], c, m, 100 * m / tot, f / m / 1e9 if m else 0, b / m / 1e6 if m else 0))

# ---- optional: tile configurations of the LDS-DMA 3x3 convolutions (csrc/conv2.hip tuning overrides) on this plan's shapes ----------------
#     python tools/arch_kernels.py hourglass --codes 0 0x404 0x204 0x104 0x3404 0x208 0x108 0x3108
if getattr(args, "codes", None):
    codes = [int(c, 0) for c in args.codes]
    ops = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV_STREAM and op.stride == 1]
    shapes = sorted({(plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) for i in ops}, key=lambda s: -s[2])
    one = (ctypes.c_float * 1)()
    print("%-26s" % "Cin->Cout @H (n)" + "".join("%10s" % ("auto" if c == 0 else hex(c)) for c in codes))
    for sh in shapes:
        idx = [i for i in ops if (plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].H) == sh]
        row = []
        for c in codes:
            op = _lib.H3dOp()
            ctypes.memmove(ctypes.byref(op), ctypes.byref(plan.op_array[idx[0]]), ctypes.sizeof(_lib.H3dOp))
            op.reserved = c
            arr = (_lib.H3dOp * 1)(op)
            ts = []
            for _ in range(args.reps + 2):
                if _lib.lib().h3d_run_ops_timed(arr, 1, _lib.stream_ptr(), one):
                    ts = [float("nan")] * 3
                    break
                ts.append(one[0])
            row.append(float(np.median(ts[2:])) * len(idx))
        print("%-26s" % ("%d->%d @%d (%d)" % (sh + (len(idx),))) + "".join("%10.4f" % v for v in row))
