"""Profiling aid (ABLATE build only): per workgroup, the share of the fused-heads kernel that wave 0 spends in the stage
barriers (waiting for the slowest wave and for the next stage's LDS-DMA) and in the 1x1 contraction (gemm2).
    make -C human-3d-reconstruction_amd/csrc ABLATE=1 && python tools/stamp_heads.py"""
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
DBG = int([a for a in sys.argv if a.startswith("dbg=")][0][4:]) if any(a.startswith("dbg=") for a in sys.argv) else 0
for i in [i for i, op in enumerate(plan.ops) if op.kind == _lib.OP_HEADS]:
    op = plan.ops[i]
    arr = (_lib.H3dOp * 1)(op)
    arr[0].reserved = DBG          # ablation bits: 1 no weight stream, 2 no stage barrier, 4 / 8 only the older / younger wave of a SIMD computes
    for _ in range(2):
        _lib.check(L.h3d_run_ops(arr, 1, _lib.stream_ptr()), "run")
    torch.cuda.synchronize()
    n = min(op.B * ((op.H + 15) // 16) * ((op.W + 31) // 32), 65536)
    buf = np.zeros(n * 8, dtype=np.uint64)
    assert L.h3d_debug_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8).astype(np.int64)
    tot = (t[:, 1] - t[:, 0]).astype(np.float64)
    if "--waves" in sys.argv:
        arr[0].reserved = 64
        _lib.check(L.h3d_run_ops(arr, 1, _lib.stream_ptr()), "run")
        torch.cuda.synchronize()
        assert L.h3d_debug_stamps(buf.ctypes.data, n * 8) == 0
        tw = buf.reshape(n, 8).astype(np.float64)
        print("   barrier share per wave 0..7: " + " ".join("%.1f%%" % (100 * (tw[:, w] / tot).mean()) for w in range(8)))
    print("op %d %s: %d workgroups, mean %.0f ticks; tap-row stages (fragment reads + MFMA) %.1f %%, own DMA pieces (vmcnt) %.1f %%, barrier %.1f %%, gemm2 %.1f %%" % (
        i, kernel_name(op).replace("unsigned short", "bf"), n, tot.mean(), 100 * (t[:, 5] / tot).mean(), 100 * (t[:, 4] / tot).mean(), 100 * (t[:, 2] / tot).mean(), 100 * (t[:, 3] / tot).mean()))
