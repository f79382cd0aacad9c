#!/usr/bin/env python
"""The rule of DLAEngine.calibrate_dcn_margins against the stopwatch it replaced: per fused DeformConv layer of the batch-64
plan, the rule's cost model (narrow / slots512 / wide, in units of a normal tile) beside HIP-event times of the three variants
(DLAEngine.time_dcn_variants), at several --offset-scale values; prints where the two disagree by more than the 3 % the rule
asks for and what the disagreement costs.  Usage (GPU box): python tools/fit_dcn_rule.py [offset_scale ...]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import h3d_amd  # noqa: E402,F401
from h3d_amd import arch, model, synth  # noqa: E402

HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2, "pose": 72, "shape": 10}
scales = [float(a) for a in sys.argv[1:]] or [0.5, 1.0, 2.0]
dev = torch.device("cuda:0")
images = torch.from_numpy(synth.synth_image_batch(64, 512, 512, seed=317)).to(dev)
for sc in scales:
    sd = synth.synth_state_dict(arch.state_dict_shapes(HEADS, True), seed=0, gain=1.25, offset_scale=sc)
    m = model.dla_net(HEADS, dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.to(dev).eval()
    eng = m.engine(dev)
    rep = eng.calibrate_dcn_margins(images)
    eng.pw.dcn_variant = {}
    eng.plans.clear()
    times = eng.time_dcn_variants(images, reps=5)
    t_rule = t_best = t_narrow = 0.0
    for p, r in rep.items():
        t = times[p]
        best = min(t, key=t.get)
        t_rule += t[r["choice"]]
        t_best += t[best]
        t_narrow += t["narrow"]
        flag = "" if t[r["choice"]] <= 1.03 * t[best] else "   <-- rule %s, stopwatch %s (+%.1f %%)" % (r["choice"], best, 100 * (t[r["choice"]] / t[best] - 1))
        print("scale %.1f %-22s rule %-8s cost %s  over256 %.3f over512 %.3f wide256 %.3f | ms %s%s"
              % (sc, p, r["choice"], json.dumps(r["cost"]), r["tiles_over_256"], r["tiles_over_512"], r["tiles_over_256_wide"],
                 json.dumps({k: round(v, 4) for k, v in t.items()}), flag))
    print("scale %.1f: DeformConv ms per step: all narrow %.3f, rule %.3f, per-layer stopwatch best %.3f" % (sc, t_narrow, t_rule, t_best))
    del m, eng
    torch.cuda.empty_cache()
