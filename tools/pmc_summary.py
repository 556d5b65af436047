#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name (short)."""
import csv, sys, collections, glob, os, re
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                m = re.search(r"(bpr_\w+|eval_\w+|rank_metrics\w*|shard_\w+|ncf_\w+|fm_\w+|cml_\w+|rows_\w+|apply_all|small_adam\w*|topk_\w+)(<[^>(]*>)?", name)
                if not m:
                    continue
                key = (m.group(1) + (m.group(2) or "").replace(" ", ""), row["Counter_Name"])
                acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            print("%-30s %-26s mean %16.1f  n=%d" % (k, c, s / n, n))
