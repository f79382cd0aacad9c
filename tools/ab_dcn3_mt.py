"""Profiling aid: 128-channel workgroups (default) vs 64-channel workgroups (h3d_op.reserved = 0x200) on the register-staged
fused DeformConvs of the bench plan, in ONE process."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
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
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_DCN_FUSED and op.Cout > 64]
res = {}
for rep in range(3):
    for cfg in (0, 0x200, 0x400):
        for i in idx:
            plan.op_array[i].reserved = cfg
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
for i in idx:
    plan.op_array[i].reserved = 0
print("op (Cin,Cout,H): default  64-ch workgroups  128-ch workgroups")
for i in idx:
    op = plan.ops[i]
    print(i, (op.Cin, op.Cout, op.H), " ".join("%.3f" % res[c][i] for c in (0, 0x200, 0x400)))
