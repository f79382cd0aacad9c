"""Profiling aid (ABLATE build only): in-kernel phase stamps of one csrc/dcn3.hip launch of the bench plan.
    make -C human-3d-reconstruction_amd/csrc ABLATE=1 && python tools/stamp_dcn.py [op index] [gain]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
STAGE_B = "--stage-b" in sys.argv      # library built with ABLATE=1 and -DDCN3_STAMP_B: sub-steps of phase B's second stage
DTYPE = "f16x3" if "--f16x3" in sys.argv else "bf16"
sys.argv = [a for a in sys.argv if a not in ("--stage-b", "--f16x3")]
dev = torch.device("cuda:0")
opt = Opt(input_h=512, input_w=512, smpl=True, dtype=DTYPE)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=float(sys.argv[2]) if len(sys.argv) > 2 else 1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
B = 64
x = torch.from_numpy(synth.synth_images(1, 512, 512)).to(dev).expand(B, 3, 512, 512).contiguous()
det.run(x); torch.cuda.synchronize()
plan = det.model.engine(dev).plan(B, 512, 512)
L = _lib.lib()
L.h3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["start->zero-filled", "phase A", "geometry+list", "phase B", "pass 2", "epilogue"]
if STAGE_B:
    names = ["patch loads issued + barrier 1", "apron store + patch commit + vmcnt(0)", "barrier 2", "filter DMA + apron loads issued", "gather/blend/MFMA"]
for i in ([int(sys.argv[1])] if len(sys.argv) > 1 and int(sys.argv[1]) >= 0 else [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_DCN_FUSED_STREAM]):
    op = plan.ops[i]
    arr = (_lib.H3dOp * 1)(op)
    for _ in range(2):
        _lib.check(L.h3d_run_ops(arr, 1, _lib.stream_ptr()), "run")
    torch.cuda.synchronize()
    nwg = op.B * ((op.H + 15) // 16) * ((op.W + 15) // 16)
    n = min(nwg, 65536)
    buf = np.zeros(n * 8, dtype=np.uint64)
    assert L.h3d_debug_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8).astype(np.int64)
    order = [0, 1, 2, 3, 4, 5] if STAGE_B else [6, 0, 1, 2, 3, 4, 5]
    d = np.stack([t[:, order[k + 1]] - t[:, order[k]] for k in range(len(order) - 1)], 1)
    tot = t[:, 5] - t[:, order[0]]
    print("op %d %s Cin=%d Cout=%d %dx%d: %d workgroups, mean %d clocks per tile (100 MHz ticks x?), span %d" % (
        i, kernel_name(op).replace("unsigned short", "bf"), op.Cin, op.Cout, op.H, op.W, n, tot.mean(), t[:, 5].max() - t[:, 6].min()))
    print("   " + "  ".join("%s %.1f%%" % (names[k], 100.0 * d[:, k].mean() / tot.mean()) for k in range(d.shape[1])))
