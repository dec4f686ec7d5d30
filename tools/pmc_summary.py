#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into HBM bytes per decode step.

Usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <steps incl. warmup> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read (MI355X_MICROARCH.md "HBM"), which is the access pattern of every
weight load here, so the fetch side is doubled; WRITE_SIZE is exact for the stores used.
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if k.startswith("__amd_rocclr"):
            continue            # load-time uploads / memsets, not part of a decode step
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    fetch, write, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = per_kernel(fetch), per_kernel(write)
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1])):
        calls = f.get(k, w.get(k))[0]
        fb = 2.0 * f.get(k, [0, 0.0])[1] * 1024 / steps
        wb = w.get(k, [0, 0.0])[1] * 1024 / steps
        rows.append({"kernel": k, "launches_per_step": calls / steps, "fetch_bytes_per_step": round(fb), "write_bytes_per_step": round(wb)})
    total = sum(r["fetch_bytes_per_step"] + r["write_bytes_per_step"] for r in rows)
    json.dump({"steps": steps, "hbm_bytes_per_step": total, "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1, KiB units",
               "kernels": rows}, open(out, "w"), indent=1)
    print(f"HBM bytes per decode step: {total / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
