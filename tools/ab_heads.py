"""Profiling aid: heads launches grouped per output-tile count (default) vs wide heads merged."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
B = 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
for wide in (0, 3):
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
    eng = det.model.engine(dev)
    eng.wide_heads_m2 = wide
    det.run(x); torch.cuda.synchronize()
    plan = eng.plan(B, 512, 512)
    n = len(plan.ops)
    ms = (ctypes.c_float * n)()
    tot = np.zeros(n)
    for _ in range(5):
        _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
        tot += np.frombuffer(ms, dtype=np.float32, count=n)
    tot /= 5
    idx = [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_HEADS]
    print("wide=%d:" % wide, " ".join("%s %.3f" % (kernel_name(plan.ops[i])[-6:], tot[i]) for i in idx), "sum %.3f" % tot[idx].sum())
