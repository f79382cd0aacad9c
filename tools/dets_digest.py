#!/usr/bin/env python
"""One fresh process: build the bf16 DLA-34 detector on synthetic weights (given --offset-scale), choose the DeformConv tile
variants with the deterministic rule, run `batch` synthetic images and print sha256(dets | inds | every head) -- two processes
must print the same line (tests/test_gpu_network.py::test_two_processes_return_identical_bits)."""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import h3d_amd  # noqa: E402,F401
from h3d_amd import arch, synth  # noqa: E402
from h3d_amd.detector import MultiPoseDetector, Opt  # noqa: E402

scale, batch, size = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
opt = Opt(input_h=size, input_w=size, dtype="bf16", K=100)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25, offset_scale=scale)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
x = torch.from_numpy(synth.synth_image_batch(batch, size, size, seed=317)).to(dev)
eng = det.model.engine(dev)
eng.calibrate_dcn_margins(x)
res = det.run(x)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in [res["dets"], res["inds"]] + [res["heads"][k] for k in sorted(res["heads"])]:
    h.update(t.cpu().numpy().tobytes())
print("DIGEST %s variants %s" % (h.hexdigest(), sorted(eng.pw.dcn_variant.items())))
