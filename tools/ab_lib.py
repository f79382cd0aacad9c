"""Profiling aid: A/B two BUILDS of libh3d_hip.so inside ONE process (boxes differ by several percent, so only same-run
comparisons count): the bench plan is built with the in-tree library, then the same `h3d_op` array is timed launch by launch
(`h3d_run_ops_timed`) through the in-tree library and through every other build given on the command line, alternating.

    cp human-3d-reconstruction_amd/csrc/libh3d_hip.so exp/lib_A.so      # the baseline build (exp/ is git-ignored, travels to the GPU box)
    ... edit, make ...
    python tools/ab_lib.py exp/lib_A.so [--batch 64] [--dtype bf16] [--offset-scale 0.5] [--filter dcn3]
"""
import argparse, ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name, op_bytes

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--offset-scale", type=float, default=0.5)
ap.add_argument("--filter", default="")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--per-op", action="store_true", help="one line per launch shape (Cin, Cout, H) instead of one per kernel name")
args = ap.parse_args()
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype=args.dtype)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25, offset_scale=args.offset_scale)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_image_batch(args.batch, 512, 512)).to(dev)
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(args.batch, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
names = [kernel_name(op) for op in plan.ops]
builds = {"in-tree": _lib.lib()}
for p in args.libs:
    L = ctypes.CDLL(p)
    L.h3d_abi_version.restype = ctypes.c_int
    if L.h3d_abi_version() != _lib.ABI_VERSION:      # (ABI 2 appended two fields to h3d_op: an older build would walk the op array with the wrong stride)
        raise SystemExit("%s: ABI %d, this tree binds ABI %d -- rebuild the baseline from the same include/h3d.h" % (p, L.h3d_abi_version(), _lib.ABI_VERSION))
    L.h3d_run_ops_timed.argtypes = _lib.lib().h3d_run_ops_timed.argtypes
    L.h3d_run_ops_timed.restype = ctypes.c_int
    builds[p] = L
runs = {k: [] for k in builds}
for rep in range(args.reps + 1):
    for k, L in builds.items():
        rc = L.h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms)
        if rc:
            raise RuntimeError("%s: h3d_run_ops_timed rc %d" % (k, rc))
        if rep:
            runs[k].append(np.frombuffer(ms, dtype=np.float32, count=n).copy())
med = {k: np.median(np.stack(v), axis=0) for k, v in runs.items()}
fam = {}
for i, nm in enumerate(names):
    if args.filter in nm:
        op = plan.ops[i]
        fam.setdefault(nm + (" %d->%d @%d" % (op.Cin, op.Cout, op.H) if args.per_op else ""), []).append(i)
keys = list(builds)
print("%-70s %4s " % ("kernel", "n") + " ".join("%12s" % k[-12:] for k in keys) + "   ratio(last/first)")
for nm, idx in sorted(fam.items(), key=lambda kv: -med[keys[0]][kv[1]].sum()):
    t = [med[k][idx].sum() for k in keys]
    gb = sum(op_bytes(plan.ops[i]) for i in idx) / 1e9       # algorithmic HBM bytes of these launches
    print("%-70s %4d " % (nm[-70:], len(idx)) + " ".join("%12.4f" % v for v in t) + "   %.3f   %6.0f GB/s" % (t[-1] / t[0], gb / (t[0] * 1e-3)))
tot = [sum(med[k][i] for idx in fam.values() for i in idx) for k in keys]
print("%-70s %4s " % ("total", "") + " ".join("%12.4f" % v for v in tot) + "   %.3f" % (tot[-1] / tot[0]))
