"""Profiling aid: run only the DCN (and conv) ops of the bench plan a few times (for rocprofv3 --pmc)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=False, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
for _ in range(2):
    det.model(x)
torch.cuda.synchronize()
