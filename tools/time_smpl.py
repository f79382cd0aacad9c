"""Profiling aid: time the SMPL kernels at the bench size (6400 persons).  With a `make ABLATE=1` build,
H3D_SMPL_ABLATE=<bits> changes smpl_verts3: bit 0 stops after the contraction; bits 1-2 select the contraction mode
(1 no MFMAs, 2 no direction-fragment reads, 3 no DMA after stage 0), e.g. 3 = contraction only, without MFMAs."""
import sys, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import smpl, synth
dev = torch.device("cuda:0")
model = smpl.SMPLModel.synthetic(seed=0)
P = 6400
betas = torch.from_numpy(synth.normalish("b", (P, 10), 0, 1, 3)).to(dev)
thetas = torch.from_numpy(synth.normalish("t", (P, 72), 0, 0.2, 3)).to(dev)
for kern in ("gen2", "gen3"):
    for _ in range(3):
        smpl.lbs(model, betas, thetas, kernel=kern)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        smpl.lbs(model, betas, thetas, kernel=kern)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.3f ms per call (pose + verts, incl. host allocs)" % (kern, e0.elapsed_time(e1) / 10))
