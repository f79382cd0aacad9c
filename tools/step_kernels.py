"""Profiling aid: run ONLY the timed step of bench.py (det.run on one stream) N times -- under
`rocprofv3 --kernel-trace --stats` this lists exactly the kernels of a step (no setup / roofline / baseline work)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, smpl_people=100, dtype="bf16", K=100)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(64, 3, 512, 512).contiguous()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    det.run(x)
torch.cuda.synchronize()
