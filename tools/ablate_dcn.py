"""Profiling aid: time the DCN ops of the bench plan with the kernel's ablation switches."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name, op_flops
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=float(sys.argv[3]) if len(sys.argv) > 3 else 1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
KIND = {'dcn': _lib.OP_DCN, 'heads': _lib.OP_HEADS, 'dcnf': _lib.OP_DCN_FUSED, 'dcns': _lib.OP_DCN_FUSED_STREAM, 'dcn4': _lib.OP_DCN_FUSED_F16, 'updcn': _lib.OP_UPDCN_F16}[sys.argv[2] if len(sys.argv) > 2 else 'dcn']
idx = [i for i, op in enumerate(plan.ops) if op.kind == KIND]
keep = {i: plan.ops[i].reserved & 0x58600 for i in idx}      # variant bits (margin / slots / workgroup width / fp16 input) stay: only the low ablation bits change
for dbg in ((0, 1) if KIND == _lib.OP_HEADS else (0, 1, 2, 3, 4) if KIND == _lib.OP_UPDCN_F16 else (0, 0x100, 1, 2, 3, 4, 0x101, 0x102, 0x103, 0x104) if KIND == _lib.OP_DCN_FUSED_F16 else (0, 1, 2, 4, 6, 7, 8, 15, 16, 31)):
    for i in idx:
        plan.op_array[i].reserved = dbg | (keep[i] if KIND == _lib.OP_DCN_FUSED_STREAM else 0)
    tot = np.zeros(n)
    for _ in range(3):
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
        tot += np.frombuffer(ms, dtype=np.float32, count=n)
    tot /= 3
    print("dbg=%d" % dbg, " ".join("%s:%.3f" % (kernel_name(plan.ops[i])[:40].replace("unsigned short", "bf"), tot[i]) for i in idx[:16]))
