"""Profiling aid: A/B the tile configurations of the 1x1 (root / project) convolutions of the bench plan in ONE process."""
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
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV and op.ksize == 1 and op.Cin % 64 == 0 and op.Cout > 32]
cfgs = [0, 0x1000 | 4 << 4 | 2, 0x1000 | 4 << 4 | 1, 0x1000 | 2 << 4 | 2, 0x1000 | 2 << 4 | 1]
res = {}
for rep in range(3):
    for cfg in cfgs:
        for i in idx:
            plan.op_array[i].reserved = cfg if (cfg == 0 or plan.ops[i].Cout >= 32 * ((cfg >> 4) & 15)) else 0
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
for i in idx:
    plan.op_array[i].reserved = 0
print("op (Cin,Cout,H): default  <4,16> <4,8> <2,16> <2,8>")
for i in idx:
    op = plan.ops[i]
    print(i, (op.Cin, op.Cout, op.H), " ".join("%.3f" % res[c][i] for c in cfgs))
