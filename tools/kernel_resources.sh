#!/bin/bash
# Per-kernel register / scratch / LDS usage of one csrc source, as hipcc's resource-usage remarks report it:
#   tools/kernel_resources.sh dcn3.hip [extra hipcc flags]
# (device-only compile, nothing is written; use it to check that a change did not push a variant over its VGPR cap or into scratch)
src=$1; shift
cd "$(dirname "$0")/../human-3d-reconstruction_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re, sys
cur = None; rows = []
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None and key not in cur: cur[key] = int(m.group(1))
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", name); name = re.sub(r"\(.*\)$", "", name)
    print("%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (name[:70], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))
'
