"""Pretty-print the per-kernel table of a bench.py JSON line (profiling aid; `--md` prints the markdown table DESIGN.md uses).
Each kernel family: device time per step, launches, algorithmic TFLOP/s and its share of the dense bf16 MFMA peak
(2500), algorithmic HBM GB/s and its share of 8000; the larger share names the roofline that bounds the family."""
import json, sys
md = "--md" in sys.argv
path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
print(d["value"], "img/s", d["ms_per_step"], "ms/step")
if md:
    print("| kernel family | ms / step | launches | TFLOP/s (of 2500) | GB/s algorithmic (of 8000) | bound |")
    print("|---|---|---|---|---|---|")
for k, v in d.get("kernels", {}).items():
    tf, gb = v.get("tflops", 0.0), v.get("gbs", 0.0)
    ff, fb = tf / 2500.0, gb / 8000.0
    bound = "MFMA" if ff >= fb else "HBM"
    name = k.replace("unsigned short", "bf16")
    if md:
        print("| `%s` | %.3f | %d | %.0f (%.2f) | %.0f (%.2f) | %s |" % (name, v["ms"], v["n"], tf, ff, gb, fb, bound))
    else:
        print(f"{v['ms']:7.3f} {v['n']:3d} {tf:7.1f} TF/s {ff:5.2f}  {gb:6.0f} GB/s {fb:5.2f}  {bound:4s} {name}")
