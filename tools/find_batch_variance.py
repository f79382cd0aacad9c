"""Debug aid: feed ABAB... (two images repeated) through a plan and report, in plan order, the ops whose output differs between
repeats of the same image (a tile-indexing bug or a race shows up at the first such op).
    python tools/find_batch_variance.py [batch] [dtype] [runs]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import h3d_amd
from h3d_amd import _lib, arch, synth
from h3d_amd.detector import MultiPoseDetector, Opt
from bench import kernel_name
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
opt = Opt(input_h=512, input_w=512, smpl=True, dtype=dtype)
sd = synth.synth_state_dict(arch.state_dict_shapes(opt.heads, True), seed=0, gain=1.25)
det = MultiPoseDetector(opt, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=dev)
two = torch.from_numpy(synth.synth_images(2, 512, 512, seed=317)).to(dev)
x = two.repeat(B // 2, 1, 1, 1).contiguous()
eng = det.model.engine(dev)
plan = eng.plan(B, 512, 512)
bufs = [t for t in plan.keep if torch.is_tensor(t) and t.dim() == 4 and t.shape[0] == B]
outs = list(plan.outputs.values())
for run in range(runs):
    det.model(x)
    torch.cuda.synchronize()
    bad = []
    for i, op in enumerate(plan.ops):
        if op.kind == _lib.OP_HEADS:
            tgt = outs
        else:
            tgt = [t for t in bufs if op.out and t.data_ptr() <= op.out < t.data_ptr() + t.numel() * t.element_size()]
        for t in tgt:
          r = t.view(B // 2, 2, *t.shape[1:])
          if not torch.equal(r, r[:1].expand_as(r)):
            d = (r.float() - r[:1].float()).abs()
            per = d.reshape(B // 2, -1).amax(1).tolist()
            bad.append((i, kernel_name(op), (op.Cin, op.Cout, op.H, op.W), [round(v, 4) for v in per]))
            break
    print("run %d: %d ops with batch-dependent outputs" % (run, len(bad)))
    for b in bad[:6]:
        print("   ", b)
