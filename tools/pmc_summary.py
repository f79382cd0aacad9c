"""Profiling aid: fold two rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one
`--pmc WRITE_SIZE` pass of the same bench command) into profiles/<tag>_pmc_traffic.json.

    python tools/pmc_summary.py fetch.csv write.csv profiles/r01_pmc_traffic.json "build note"

Counters are reported in KiB per dispatch; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes
for gfx950 (it reports half of a wide coalesced stream).  Kernel names are normalised the way
bench.py's `kernel_name()` prints them ("void " prefix and the argument list stripped).
"""
import csv, json, re, sys
from collections import defaultdict


def norm(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name).strip()
    return re.sub(r"(, false)+>$", ">", name)      # trailing defaulted template arguments: the launchers' names omit them


def fold(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[norm(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"]) * 1024.0
            a[1] += 1
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe[k][0] / max(fe[k][1], 1)
        w = wr[k][0] / max(wr[k][1], 1)
        kernels[k] = {"launches_sampled": max(fe[k][1], wr[k][1]),
                      "FETCH_SIZE_bytes_per_launch": int(f),
                      "fetch_bytes_x2_gfx950_correction": int(2 * f),
                      "WRITE_SIZE_bytes_per_launch": int(w),
                      "hbm_bytes_per_launch": int(2 * f + w)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench command; "
                       "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950). " + note,
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
