#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace csv: per kernel, calls, average duration, and the average idle gap between its
end and the start of the next dispatch on the device (same queue).  usage: trace_gaps.py <dir or *_kernel_trace.csv>"""
import csv
import glob
import os
import re
import sys


def main(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    stat = {}
    for k, r in enumerate(rows):
        m = re.search(r"(\w+)(<[^>]*>)?\(", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
        name = name.replace("(anonymous namespace)::", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = int(rows[k + 1]["Start_Timestamp"]) - e if k + 1 < len(rows) else 0
        a = stat.setdefault(name, [0, 0, 0, 0])
        a[0] += 1
        a[1] += e - s
        if gap < 200_000:  # ignore host-side pauses
            a[2] += gap
            a[3] += 1
    print("%-56s %8s %10s %12s" % ("kernel", "calls", "avg_us", "gap_after_us"))
    for name, (n, tot, gap, ng) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        print("%-56s %8d %10.2f %12.2f" % (name[:56], n, tot / n / 1e3, gap / max(ng, 1) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
