"""Pretty-print the per-kernel table of a bench.py JSON line (profiling aid)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], "img/s", d["ms_per_step"], "ms/step")
for k, v in d.get("kernels", {}).items():
    print(f"{v['ms']:7.3f} {v['n']:3d} {k}")
