"""TEST INFRASTRUCTURE (imports oracle/, hence it lives under tests/).  CPU aid: per-DeformConv-layer offset statistics of the synthetic weights (oracle network, fp32):
mean |offset|, tail probabilities, and the share of samples that leave a margin-M apron of a 16x16 tile.
    python tests/offset_stats.py [gain] [offset_scale] [hw]"""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import h3d_amd  # noqa
from h3d_amd import arch, synth
from h3d_amd.detector import Opt
from oracle import dla as odla

gain = float(sys.argv[1]) if len(sys.argv) > 1 else 1.25
oscale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 512
opt = Opt(smpl=True)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=gain, offset_scale=oscale)
net = odla.DLAOracle(sd, opt.heads, use_dcn=True)
rows = []
orig = net._deform_conv


def hook(x, p):
    om = F.conv2d(x, net.sd[p + ".conv.conv_offset_mask.weight"], net.sd[p + ".conv.conv_offset_mask.bias"], 1, 1)
    off = om[:, :18]
    B, _, H, W = off.shape
    ys = torch.arange(H).view(1, H, 1).float()
    xs = torch.arange(W).view(1, 1, W).float()
    fr = {}
    for M in (1, 2, 4, 6, 8):
        HH = 18 + 2 * M
        slow = tot = 0
        for t in range(9):
            ti, tj = divmod(t, 3)
            h_im, w_im = ys - 1 + ti + off[:, 2 * t], xs - 1 + tj + off[:, 2 * t + 1]
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            ry = torch.floor(h_im) - (ys - ys % 16 - 1 - M)
            rx = torch.floor(w_im) - (xs - xs % 16 - 1 - M)
            ok = (ry >= 0) & (ry + 1 < HH) & (rx >= 0) & (rx + 1 < HH)
            slow += int((inside & ~ok).sum())
            tot += inside.numel()
        fr[M] = slow / tot
    a = off.abs()
    rows.append((p, x.shape[1], x.shape[2], float(x.abs().mean()), float(a.mean()), float(a.median()), float((a > 2).float().mean()),
                 float((a > 4).float().mean()), float((a > 8).float().mean()), fr))
    return orig(x, p)


net._deform_conv = hook
with torch.no_grad():
    net(torch.from_numpy(synth.synth_images(1, hw, hw, seed=317)))
print("gain %g offset_scale %g %dx%d" % (gain, oscale, hw, hw))
print("%-22s %4s %4s %7s %7s %7s %6s %6s %6s | slow frac at margin 1 2 4 6 8" % ("layer", "Cin", "H", "|x|", "mean|o|", "med|o|", ">2", ">4", ">8"))
for r in rows:
    print("%-22s %4d %4d %7.3f %7.3f %7.3f %6.3f %6.3f %6.3f | %s" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8],
                                                                 " ".join("%.4f" % r[9][m] for m in (1, 2, 4, 6, 8))))
