#!/usr/bin/env python3
"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite file: rocprofv3 7.x writes
`*_results.db` by default; this prints/saves the per-kernel Calls / Total / Average / Min / Max in ns as CSV."""
import csv
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / total:.2f}", r[4], r[5]])


if __name__ == "__main__":
    main(*sys.argv[1:3])
