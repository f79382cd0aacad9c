"""Profiling aid: the stand-alone `DCN` module (one fused launch in the model's configuration) in its two fp32-tensor arithmetics:
split-operand fp16 MFMAs (default since round 5) and the fp32 matrix instruction (dcn_v2.OP_F32_MFMA)."""
import sys, torch
sys.path.insert(0, ".")
import h3d_amd  # noqa: F401
from h3d_amd import dcn_v2
dev = "cuda:0"
for (cin, cout, hw, B) in ((64, 64, 128, 16), (128, 128, 64, 16), (256, 256, 32, 16), (512, 256, 16, 16)):
    m = dcn_v2.DCN(cin, cout, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(dev).eval()
    with torch.no_grad():
        m.conv_offset_mask.weight.copy_(torch.randn_like(m.conv_offset_mask.weight) * 0.02)
    x = torch.randn(B, cin, hw, hw, device=dev)
    res = {}
    for name, flag in (("f16x3", False), ("f32_mfma", True)):
        dcn_v2.OP_F32_MFMA = flag
        for _ in range(3):
            m(x)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m(x); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[name] = best
    dcn_v2.OP_F32_MFMA = False
    print("DCN(%d -> %d) on [%d, %d, %d, %d]: f16x3 %.3f ms, fp32 MFMA %.3f ms" % (cin, cout, B, cin, hw, hw, res["f16x3"], res["f32_mfma"]))
