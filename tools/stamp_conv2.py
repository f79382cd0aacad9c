"""Profiling aid (ABLATE build only): in the K loop of one csrc/conv2.hip launch of the bench plan, the share of the loop that
wave 0 spends issuing the next stage's LDS-DMA, and the share the oldest (0-3) and youngest (last three) waves of a workgroup
spend at the stage barrier.    make -C human-3d-reconstruction_amd/csrc ABLATE=1 && python tools/stamp_conv2.py [op index ...]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype="bf16")
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
L = _lib.lib()
L.h3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
ops = [int(a) for a in sys.argv[1:]] or [11, 23, 35, 4]
for i in ops:
    op = plan.ops[i]
    if op.kind != _lib.OP_CONV_STREAM:
        continue
    arr = (_lib.H3dOp * 1)(op)
    for _ in range(2):
        _lib.check(L.h3d_run_ops(arr, 1, _lib.stream_ptr()), "run")
    torch.cuda.synchronize()
    n = 256
    buf = np.zeros(n * 8, dtype=np.uint64)
    assert L.h3d_debug_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8)
    loop = (t[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
    iss = (t[:, 7] >> np.uint64(32)).astype(np.float64)
    bar = t[:, :7].astype(np.float64)
    print("op %d %s Cin=%d Cout=%d %dx%d: K loop %.0f ticks; wave 0 issues DMA %.1f %%; barrier wait, waves 0-3: %s  three youngest: %s" % (
        i, kernel_name(op), op.Cin, op.Cout, op.H, op.W, loop.mean(), 100 * (iss / loop).mean(),
        " ".join("%.0f%%" % (100 * (bar[:, w] / loop).mean()) for w in range(4)),
        " ".join("%.0f%%" % (100 * (bar[:, w] / loop).mean()) for w in range(4, 7))))
