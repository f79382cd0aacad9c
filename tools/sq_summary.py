"""Profiling aid: fold rocprofv3 SQ counter passes (tools/profile_round.sh: pmc_sq1, pmc_sq2) into a per-kernel table.

    python tools/sq_summary.py gpurun_out/pmc_sq1/q_counter_collection.csv gpurun_out/pmc_sq2/q_counter_collection.csv \
        profiles/r01_pmc_sq_summary.json

Per kernel (mean over its dispatches):
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed
               over the 8 XCDs: 24.5M for a 1.40 ms launch = 8 x 2.19 GHz)   (MFMA pipe busy share, all SIMDs)
  wait_any / wait_inst / active = share of SQ_WAVE_CYCLES a wave is parked (s_waitcnt, barrier) / stalled at
               issue / issuing (MI355X_MICROARCH.md 'rocprofv3 PMC slots': the three are disjoint)
  valu_per_mfma, lds_per_mfma = instruction mix; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv, json, re, sys
from collections import defaultdict


def norm(name):
    name = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name)).strip()
    return re.sub(r"(, false)+>$", ">", name)      # trailing defaulted template arguments: the launchers' names omit them


def fold(path):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = norm(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}


def main():
    a, b, out = fold(sys.argv[1]), fold(sys.argv[2]), sys.argv[3]
    res = {}
    for k in sorted(set(a) | set(b)):
        if "at::" in k or "rocclr" in k or not k:
            continue
        x, y = a.get(k, {}), b.get(k, {})
        wc = x.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        gui = (x.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) or 1.0
        mf = y.get("SQ_INSTS_MFMA", 0.0)
        res[k] = {
            "gpu_cycles": round(gui),
            "mfma_util": round(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0), 4),
            "wait_any": round(x.get("SQ_WAIT_ANY", 0.0) / wc, 3),
            "wait_inst": round(x.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
            "wait_inst_lds": round(x.get("SQ_WAIT_INST_LDS", 0.0) / wc, 3),
            "active": round(x.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
            "insts_mfma": round(mf), "insts_valu": round(y.get("SQ_INSTS_VALU", 0.0)),
            "insts_lds": round(y.get("SQ_INSTS_LDS", 0.0)), "insts_salu": round(y.get("SQ_INSTS_SALU", 0.0)),
            "valu_per_mfma": round((y.get("SQ_INSTS_VALU", 0.0) - mf) / mf, 2) if mf else None,
            "lds_per_mfma": round(y.get("SQ_INSTS_LDS", 0.0) / mf, 2) if mf else None,
            "lds_conflict": round(y.get("SQ_LDS_BANK_CONFLICT", 0.0) / (y.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0), 3),
        }
    json.dump({"note": "rocprofv3 --pmc SQ passes of bench.py (tools/profile_round.sh); per-kernel means", "kernels": res},
              open(out, "w"), indent=1)
    print("%-52s %9s %6s %6s %6s %6s %7s %7s %6s" % ("kernel", "cycles", "mfma", "park", "stall", "issue", "valu/mf", "lds/mf", "confl"))
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["gpu_cycles"]):
        print("%-52s %9d %6.3f %6.3f %6.3f %6.3f %7s %7s %6.3f" % (k[:52], v["gpu_cycles"], v["mfma_util"], v["wait_any"], v["wait_inst"],
                                                                  v["active"], v["valu_per_mfma"], v["lds_per_mfma"], v["lds_conflict"]))


if __name__ == "__main__":
    main()
