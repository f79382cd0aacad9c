"""Profiling aid: per-op time of the bench plan (HIP events around every op), with shapes.
Lowering flags (engine.Plan.FLAGS) can be overridden: python tools/list_ops.py fuse_upnode_min_f=2 dense_dcn3=0
Synthetic weights: gain=1.25 offset_scale=0.5 (h3d_amd.synth); batch=64"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name, op_flops
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
kw = dict(kv.split("=") for kv in sys.argv[1:])
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=float(kw.pop("gain", 1.25)),
                            offset_scale=float(kw.pop("offset_scale", 0.5)))
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = int(kw.pop("batch", 64))
for k, v in kw.items():
    setattr(det.model.engine(dev), k, int(v))
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
n = len(plan.ops)
ms = (ctypes.c_float * n)()
tot = np.zeros(n)
for _ in range(5):
    _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "timed")
    tot += np.frombuffer(ms, dtype=np.float32, count=n)
tot /= 5
for i, op in enumerate(plan.ops):
    fl = op_flops(op)
    print("%3d %7.3f ms %7.1f TF/s  Cin=%4d Cout=%4d %3dx%-3d k%d s%d  %s" % (
        i, tot[i], fl / tot[i] / 1e9 if tot[i] > 0 else 0, op.Cin, op.Cout, op.Ho, op.Wo, op.ksize, op.stride,
        kernel_name(op).replace("unsigned short", "bf")))
print("total %.3f ms" % tot.sum())
