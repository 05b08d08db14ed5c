#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; counter_collection.csv) into the matrix-core
utilisation of every kernel:  pmc_mfma.py COUNTERS.csv OUT.json "note"
    util = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 * 1024)
(rocprofiler's own MfmaUtil: busy cycles of all SIMDs over elapsed cycles x 1024 SIMDs; rocprofv3 reports GRBM_GUI_ACTIVE summed
over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back').  Also the clock each kernel ran at: GRBM_GUI_ACTIVE / 8 / duration is not
in this file (no timestamps in a counter pass), so only cycles are reported."""
import csv
import json
import sys

from pmc_traffic import short

SIMDS = 1024


def main():
    busy, active, n = {}, {}, {}
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] = busy.get(k, 0.0) + v
            n[k] = n.get(k, 0) + 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            active[k] = active.get(k, 0.0) + v
    res = {"_note": sys.argv[3] if len(sys.argv) > 3 else ""}
    tot_b = tot_a = 0.0
    for k in sorted(busy, key=lambda k: -active.get(k, 0.0)):
        a = active.get(k, 0.0) / 8.0
        if a <= 0 or a * 1e-9 < 1e-5:
            continue
        tot_b += busy[k]
        tot_a += a
        if busy[k] <= 0 and a < 2e6:
            continue
        res[k] = {"launches": n[k], "mfma_busy_cycles": round(busy[k]), "elapsed_cycles": round(a),
                  "mfma_util": round(busy[k] / (a * SIMDS), 4)}
    res["_all_kernels"] = {"mfma_busy_cycles": round(tot_b), "elapsed_cycles": round(tot_a), "mfma_util": round(tot_b / (tot_a * SIMDS), 4)}
    json.dump(res, open(sys.argv[2], "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_note"):
            print(f"{k:40s} {v.get('launches', 0):5d} launches  MFMA busy {100 * v['mfma_util']:5.1f} % of {v['elapsed_cycles'] / 1e6:9.1f} Mcycles")


if __name__ == "__main__":
    main()
