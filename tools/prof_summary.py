#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short, diff-able table (kernel names truncated)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    w.writerow([name[:90], r["Calls"], f'{int(r["TotalDurationNs"]) / 1e6:.2f}', f'{float(r["AverageNs"]) / 1e3:.2f}', r["Percentage"],
                f'{int(r["MinNs"]) / 1e3:.2f}', f'{int(r["MaxNs"]) / 1e3:.2f}'])
