"""Profiling aid: the <= 64-channel DeformConvs of the fp16 bench plan through csrc/dcn5.hip's experiment variants
(h3d_op.reserved >> 16) and through csrc/dcn3.hip's register-staged apron (0x2000), in ONE process.
    python tools/ab_dcn5.py [batch] [xp ...]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="f16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
xps = [int(v, 0) for v in sys.argv[2:]] or [0, 0x4000, 0x4000 | 1 << 16, 0x4000 | 2 << 16, 0x4000 | 17 << 16]      # 0 = csrc/dcn3.hip (default)
x = torch.from_numpy(synth.synth_image_batch(B, 512, 512)).to(dev)
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_DCN_FUSED_STREAM and 32 < op.Cout <= 64]
res = {}
for rep in range(2):
    for cfg in xps:
        for i in idx:
            plan.op_array[i].reserved = cfg
        tot = []
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot.append(np.frombuffer(ms, dtype=np.float32, count=n).copy())
        res[cfg] = np.median(np.stack(tot), axis=0)
for i in idx:
    plan.op_array[i].reserved = 0
print("op (Cin,Cout,H): " + "  ".join("%#x" % c for c in xps))
for i in idx:
    op = plan.ops[i]
    print(i, (op.Cin, op.Cout, op.H), " ".join("%.4f" % res[c][i] for c in xps))
print("sum", " ".join("%.4f" % sum(res[c][i] for i in idx) for c in xps))
print("names", [kernel_name_ for kernel_name_ in {kernel_name(plan.ops[idx[0]])}])
