"""Profiling aid: heads kernel with the bias as the first MFMA's C operand (default) vs a separate bias / zeroing
pass (h3d_op.reserved = 0x200), same process, interleaved."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
B = 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
eng = det.model.engine(dev)
det.run(x); torch.cuda.synchronize()
plan = eng.plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_HEADS]
res = {0: np.zeros(n), 0x200: np.zeros(n)}
for rep in range(6):
    for flag in (0, 0x200):
        for i in idx:
            plan.op_array[i].reserved = flag
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
        if rep:
            res[flag] += np.frombuffer(ms, dtype=np.float32, count=n)
for i in idx:
    plan.op_array[i].reserved = 0
for flag in (0, 0x200):
    print("reserved=0x%x:" % flag, " ".join("%.3f" % (res[flag][i] / 5) for i in idx), "sum %.3f" % (res[flag][idx].sum() / 5))
