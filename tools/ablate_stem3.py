"""Profiling aid (make ABLATE=1): time stem3_kernel stopped after phase 1..3 and complete."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
i0 = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_STEM3][0]
for dbg in (1, 2, 3, 0):
    plan.op_array[i0].reserved = dbg
    tot = 0.0
    for _ in range(5):
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
        tot += ms[i0]
    print("stop after phase %d: %.3f ms" % (dbg, tot / 5))
