"""Round-5 parity analysis: where is the tail of the bf16 plan's `hm` error, and which rounding point of the plan
that the CPU emulation does not model produces it?  Reads the dumps of tools/dump_heads.py.

    python tools/hm_tail.py gpurun_out/r5_heads
"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from h3d_amd import arch, synth
from h3d_amd.detector import Opt
from oracle import dla as odla

d = sys.argv[1]
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16", K=100)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
two = torch.from_numpy(synth.synth_images(2, 512, 512, seed=317))
torch.set_num_threads(8)


def stats(name, got, ref):
    e = np.abs(got - ref).ravel()
    q = np.quantile(e, [0.5, 0.99, 0.999, 0.9999])
    print("%-28s max %.4f  q99.99 %.4f  q99.9 %.4f  q99 %.4f  med %.4f  rms %.4f" % (name, e.max(), q[3], q[2], q[1], q[0], np.sqrt((e ** 2).mean())))
    return e


with torch.no_grad():
    o = odla.DLAOracle(sd, opt.heads, use_dcn=True)
    ref = {k: v.numpy() for k, v in o(two)[0].items()}
    ref_feat = o.feat.permute(0, 2, 3, 1).numpy()
    variants = {"emu bf16": dict(emulate="bf16"), "emu bf16_plan": dict(emulate="bf16_plan")}
    emus = {}
    for name, kw in variants.items():
        oe = odla.DLAOracle(sd, opt.heads, use_dcn=True, **kw)
        emus[name] = ({k: v.numpy() for k, v in oe(two)[0].items()}, oe.feat.permute(0, 2, 3, 1).numpy())
print("hm logits: min %.2f max %.2f std %.2f" % (ref["hm"].min(), ref["hm"].max(), ref["hm"].std()))
for head in ("hm", "wh", "hm_hp"):
    print("==", head)
    for name, (h, f) in emus.items():
        stats(name, h[head], ref[head])
    for f in ("f32", "f16", "bf16", "bf16_nodef16_0", "bf16_unfused_heads", "bf16_plainconv", "bf16_nostem3"):
        z = np.load("%s/%s.npz" % (d, f))
        e = stats("gpu " + f, z[head], ref[head])
        if head == "hm" and f == "bf16":
            idx = np.argsort(-e)[:8]
            print("   worst pixels:", [(int(i), round(float(e[i]), 3), round(float(ref["hm"].ravel()[i]), 2)) for i in idx])
            ee = np.abs(emus["emu bf16"][0]["hm"] - ref["hm"]).ravel()
            print("   emulation error there:", [round(float(ee[i]), 3) for i in idx])
            for en in emus:
                print("   corr(gpu err, %s err) = %.3f ; |gpu - %s|: max %.4f rms %.4f ; mean signed err gpu %.4f emu %.4f" % (
                    en, np.corrcoef((z[head] - ref[head]).ravel(), (emus[en][0]["hm"] - ref["hm"]).ravel())[0, 1], en,
                    np.abs(z[head] - emus[en][0]["hm"]).max(), np.sqrt(((z[head] - emus[en][0]["hm"]) ** 2).mean()),
                    (z[head] - ref[head]).mean(), (emus[en][0]["hm"] - ref["hm"]).mean()))
print("== feat (64 ch)")
for name, (h, f) in emus.items():
    stats(name, f, ref_feat)
for f in ("f32", "f16", "bf16", "bf16_nodef16_0"):
    z = np.load("%s/%s.npz" % (d, f))
    stats("gpu " + f, z["feat"], ref_feat)
