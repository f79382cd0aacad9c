"""Profiling aid: A/B the conv_stream (csrc/conv2.hip) tile configurations on the bench plan's layers,
inside ONE process (boxes differ by several percent, so only same-run comparisons count)."""
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
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV_STREAM]
cfgs = [int(c, 16) for c in sys.argv[1:]] or [0, 0x410, 0x408, 0x404, 0x208, 0x204]
res = {}
for rep in range(2):
    for cfg in cfgs:
        for i in idx:
            op = plan.ops[i]
            mt = ((cfg & 0xfff) >> 8)
            plan.op_array[i].reserved = cfg if (cfg == 0 or (op.Cout + 31) // 32 >= mt) else 0
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
shapes = {}
for i in idx:
    op = plan.ops[i]
    shapes.setdefault((op.Cin, op.Cout, op.H), []).append(i)
print("shape (Cin,Cout,H) x n : " + "  ".join("%#5x" % c for c in cfgs))
for k, v in shapes.items():
    print(k, "x", len(v), ":", "  ".join("%5.3f" % res[c][v].sum() for c in cfgs))
