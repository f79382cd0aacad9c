"""Profiling aid: ring configurations of the stride-2 3x3 layers on csrc/conv2.hip (same process)."""
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
ref = {k: v.clone() for k, v in det.run(x)["heads"].items()}
plan = det.model.engine(dev).plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV_STREAM and op.stride == 2]
cfgs = [0, 0x2404, 0x4408, 0x4204]      # 0 = one slot (default), 0x2404 = two slots, ...
res = {}
for rep in range(3):
    for cfg in cfgs:
        for i in idx:
            plan.op_array[i].reserved = cfg
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
        if rep == 0:
            out = det.run(x)["heads"]
            print("cfg %#x max |diff| vs default: %.3g" % (cfg, max(float((out[k] - ref[k]).abs().max()) for k in ref)))
for i in idx:
    plan.op_array[i].reserved = 0
print("op (Cin,Cout,Ho): " + " ".join("%#6x" % c for c in cfgs))
for i in idx:
    op = plan.ops[i]
    print(i, (op.Cin, op.Cout, op.Ho), " ".join("%.3f" % res[c][i] for c in cfgs))
