"""Round-5 parity aid (the `hm` error tail): run the full-size test's two images (seed 317, gain 1.25) through
several plans at batch 8 and save the heads of the first two images + the 64-channel feature map in front of the heads, for analysis
against the CPU oracle / emulation variants (tools/hm_tail.py) off the GPU box.

    python tools/dump_heads.py gpurun_out/r5_heads
"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd  # noqa: F401
from h3d_amd import arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
two = synth.synth_images(2, 512, 512, seed=317)
B = 8
xs = torch.from_numpy(two).to(dev).repeat(B // 2, 1, 1, 1).contiguous()


def run(name, dtype, **flags):
    opt = Opt(input_h=512, input_w=512, smpl=True, dtype=dtype, K=100)
    sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
    det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
    eng = det.model.engine(dev)
    for k, v in flags.items():
        setattr(eng, k, v)
    eng.plans.clear()
    res = det.run(xs)
    torch.cuda.synchronize()
    plan = eng.plan(B, 512, 512)
    feat = plan.feat.buf[:2].float().cpu().numpy()
    rec = {k: v[:2].cpu().numpy() for k, v in res["heads"].items() if k in ("hm", "wh", "reg", "hm_hp", "hp_offset")}
    rec["feat"] = feat.astype(np.float32)
    rec["inds"] = res["inds"][:2].cpu().numpy()
    np.savez_compressed(os.path.join(out, name + ".npz"), **rec)
    print(name, {k: v.shape for k, v in rec.items()}, flush=True)
    del det, res
    torch.cuda.empty_cache()


run("bf16", "bf16")
run("bf16_nodef16_0", "bf16", node_f16=False)
run("bf16_unfused_heads", "bf16", fuse_heads=False)
run("bf16_plainconv", "bf16", stream_convs=False, stream_s2=False)
run("bf16_nostem3", "bf16", fuse_stem=False)
run("f16", "f16")
run("f32", "f32")
