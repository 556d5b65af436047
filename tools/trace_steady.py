#!/usr/bin/env python3
"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace csv: mean over all launches (what --stats prints),
and median / mean once the process has warmed up (launches after the first `skip`).  The first launches of a process
run with cold caches and clocks still ramping and pull the plain average up.
usage: trace_steady.py <dir or *_kernel_trace.csv> [kernel substring] [skip]"""
import csv
import glob
import os
import sys

import numpy as np


def main(path, name="bpr_step_fused", skip=150):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    s = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
    e = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
    d = (e - s) / 1e3
    period = np.diff(s) / 1e3
    print("%s: %d launches; all: mean %.2f us; after the first %d: mean %.2f us, median %.2f us, p10 %.2f, p90 %.2f; "
          "start-to-start period after the first %d: median %.2f us" % (
              name, len(d), d.mean(), skip, d[skip:].mean(), np.median(d[skip:]), np.percentile(d[skip:], 10),
              np.percentile(d[skip:], 90), skip, np.median(period[skip:])))


if __name__ == "__main__":
    main(*(sys.argv[1:2] + sys.argv[2:3] + [int(x) for x in sys.argv[3:4]]))
