"""Profiling aid: time h3d_nms_topk phases (needs `make ABLATE=1` for the early-exit flags)."""
import sys, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib
dev = torch.device("cuda:0")
B, H, W, K = 64, 128, 128, 100
for C in (1, 17):
    heat = torch.randn(B, C, H, W, device=dev) * 2 - 3
    sc = torch.empty(B, C, K, device=dev); ind = torch.empty(B, C, K, dtype=torch.int64, device=dev)
    ys = torch.empty_like(sc); xs = torch.empty_like(sc)
    for fl in (0x101, 0x201, 0x401, 0x001):
        def run():
            _lib.check(_lib.lib().h3d_nms_topk(_lib.ptr(heat), B, C, H, W, K, fl, _lib.ptr(sc), _lib.ptr(ind), _lib.ptr(ys),
                                               _lib.ptr(xs), _lib.stream_ptr()), "nms")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print("C=%d flags=%#x  %.1f us" % (C, fl, e0.elapsed_time(e1) * 100))
