#!/usr/bin/env python3
"""HBM-side bytes per launch of one kernel from a tools/pmc_summary.py file.  FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section:
128-byte requests tallied at 64 B), so it is doubled; WRITE_SIZE is taken as is.  The two counters come from separate
--pmc passes.  usage: traffic_json.py <pmc_summary.txt> <kernel substring>"""
import json
import re
import sys


def main(path, kernel):
    vals = {}
    for line in open(path):
        if kernel in line:
            m = re.match(r"(\S+)\s+(\S+)\s+mean\s+([0-9.]+)", line.strip())
            if m:
                vals[m.group(2)] = float(m.group(3))
    out = {"kernel": kernel, "FETCH_SIZE_KiB_mean": vals.get("FETCH_SIZE"), "WRITE_SIZE_KiB_mean": vals.get("WRITE_SIZE")}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out["traffic_bytes_per_launch"] = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        out["note"] = "FETCH_SIZE x 2 + WRITE_SIZE (KiB -> bytes): the gfx950 correction of MI355X_MICROARCH.md, separate --pmc passes"
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
