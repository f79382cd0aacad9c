"""Profiling aid: the 1x1 convolutions of a plan through csrc/gemm1.hip (auto / the three tile shapes) against the halo-tile
kernel of csrc/conv.hip, in ONE process.  python tools/ab_gemm1.py [dla_34|resdcn_101|hourglass] [batch] [size]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, synth
from h3d_amd.detector import Opt, make_detector
from bench import op_flops
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"dla_34": 64, "resdcn_101": 32, "hourglass": 16}[name]
hw = int(sys.argv[3]) if len(sys.argv) > 3 else {"dla_34": 512, "resdcn_101": 768, "hourglass": 512}[name]
from h3d_amd import arch, arch_hg, arch_res
if name == "dla_34":
    opt = Opt(input_h=hw, input_w=hw, smpl=True, dtype="bf16")
    shapes, gain = arch.state_dict_shapes(opt.heads, True), 1.25
elif name == "hourglass":
    opt = Opt(arch="hourglass", input_h=hw, input_w=hw, dtype="bf16")
    shapes, gain = arch_hg.state_dict_shapes(opt.heads), 0.8
else:
    opt = Opt(arch="resdcn_101", task="ctdet", input_h=hw, input_w=hw, dtype="bf16")
    shapes, gain = arch_res.state_dict_shapes(opt.heads, 64), 0.9
sd = synth.synth_state_dict(shapes, seed=0, gain=gain)
det = make_detector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_images(1, hw, hw)).to(dev).expand(B, 3, hw, hw).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, hw, hw)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV and op.ksize == 1 and op.stride == 1 and op.Cin % 64 == 0 and op.Cin >= 128]
cfgs = [0x2000, 0, 0x4100, 0x4200, 0x4300]
idx_s2 = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_CONV and op.ksize == 1 and op.stride == 2 and op.Cin % 64 == 0 and op.Cin >= 128]
idx = idx + idx_s2
res = {}
for rep in range(2):
    for cfg in cfgs:
        for i in idx:      # a forced tile needs cdiv(Cout, BM) * BM packed rows (the launcher refuses otherwise)
            bm = 256 if cfg == 0x4100 else 128
            plan.op_array[i].reserved = cfg if -(-plan.ops[i].Cout // bm) * bm <= plan.ops[i].wrows or cfg in (0, 0x2000) else 0
        tot = np.zeros(n)
        for _ in range(3):
            _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
            tot += np.frombuffer(ms, dtype=np.float32, count=n)
        res[cfg] = tot / 3
for i in idx:
    plan.op_array[i].reserved = 0
print("op (Cin,Cout,HxW,res): halo-tile | auto 256x256 128x256(3 slots) 128x128(4 waves, 2 per CU)   [TF/s of auto]")
seen = {}
for i in idx:
    op = plan.ops[i]
    key = (op.Cin, op.Cout, op.H, op.W, bool(op.in2))
    seen.setdefault(key, []).append(i)
for key, ii in seen.items():
    t = {c: float(np.mean([res[c][i] for i in ii])) for c in cfgs}
    print("%2d x %-28s %.3f | %s   [%.0f]" % (len(ii), key, t[0x2000], " ".join("%.3f" % t[c] for c in cfgs[1:]),
                                              op_flops(plan.ops[ii[0]]) / t[0] / 1e9))
print("total ms: " + "  ".join("%#x %.3f" % (c, sum(res[c][i] for i in idx)) for c in cfgs))
