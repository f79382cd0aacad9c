import sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from oracle import dla as odla
DEV="cuda:0"
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16", K=100)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
two = synth.synth_images(2, 512, 512, seed=317)
torch.set_num_threads(16)
with torch.no_grad():
    ref = {k: v.numpy() for k, v in odla.DLAOracle(sd, opt.heads, use_dcn=True)(torch.from_numpy(two))[0].items()}
    emu = {k: v.numpy() for k, v in odla.DLAOracle(sd, opt.heads, use_dcn=True, emulate_bf16=True)(torch.from_numpy(two))[0].items()}
print("emu ", {k: (round(float(np.abs(emu[k]-ref[k]).max()),3), round(float(np.sqrt(np.mean((emu[k]-ref[k])**2))),4), round(float(np.quantile(np.abs(emu[k]-ref[k]),0.9999)),3)) for k in ref})
for flag in (0, 1):
    for batch in (8, 64):
        det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=DEV)
        det.model.engine(torch.device(DEV)).node_f16 = flag
        xs = torch.from_numpy(two).to(DEV).repeat(batch // 2, 1, 1, 1).contiguous()
        res = det.run(xs)
        got = {k: v[:2].cpu().numpy() for k, v in res["heads"].items()}
        print("node_f16", flag, "batch", batch, {k: (round(float(np.abs(got[k]-ref[k]).max()),3), round(float(np.sqrt(np.mean((got[k]-ref[k])**2))),4), round(float(np.quantile(np.abs(got[k]-ref[k]),0.9999)),3)) for k in ref})
        del det, res
        torch.cuda.empty_cache()
