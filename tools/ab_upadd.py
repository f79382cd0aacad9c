"""Profiling aid: per-launch time of the up-sample+add ops with the tap table in LDS (default) or read from global."""
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
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_UPADD]
res = {}
for rep in range(2):
    for cfg in (0, 1):
        for i in idx:
            plan.op_array[i].reserved = cfg
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
for i in idx:
    op = plan.ops[i]
    print("C=%d f=%d out %dx%d f16=%d : lds %.3f  global %.3f ms" % (op.Cin, op.stride, op.Ho, op.Wo, op.out_mode == 3, res[0][i], res[1][i]))
