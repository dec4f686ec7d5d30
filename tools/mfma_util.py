#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc run (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F16).

Usage: tools/mfma_util.py <counter_collection.csv> <out.json>
  util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 * 1024)
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so a dispatch's busy window is a
eighth of it; 1024 = 256 CUs x 4 SIMDs, each with one matrix pipe; SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is
busy.  MFMA FLOP = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512."""
import collections
import csv
import json
import sys


def main(path, out):
    k = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        k[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[name] += 1
    rows = []
    for name, c in k.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        rows.append({"kernel": name[:120], "dispatches": n[name], "mfma_busy_cycles": busy, "gui_active_sum_over_xcds": gui,
                     "mfma_util_percent": round(100.0 * busy / (gui / 8.0 * 1024.0), 2),
                     "mfma_flop": c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512.0})
    rows.sort(key=lambda r: -r["mfma_busy_cycles"])
    json.dump({"formula": "100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "kernels": rows}, open(out, "w"), indent=1)
    for r in rows[:8]:
        print(f"{r['mfma_util_percent']:6.2f} %  {r['dispatches']:5d}  {r['kernel'][:90]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
