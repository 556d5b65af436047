#!/bin/bash
# per-kernel times of the device sampler (rocprofv3 --kernel-trace --stats of tools/prof_dsampler.py) -> gpurun_out/dsampler_kernel_stats.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/ds_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ds_trace -- python3 $ROOT/tools/prof_dsampler.py ${1:-pairwise} ${2:-10} > $OUT/ds_under_trace.log 2>&1
f=$(find $OUT/ds_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/dsampler_kernel_stats.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python3 tools/prof_dsampler.py (S-ml1m, neg_ratio 4: 2.8 M triples per sample)")
print("%-44s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows[:24]:
    n = r["Name"]
    m = re.search(r"(neg_\w+(<\d+>)?|perm_\w+|mt_\w+|cand_\w+|build_bitmaps|radix_sort_onesweep_\w+|__amd_rocclr_\w+)", n)
    name = (m.group(0) if m else n.split("(")[0])[:44]
    print("%-44s %8s %14s %12.1f %8s" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
PY
rm -rf $OUT/ds_trace
cat $OUT/dsampler_kernel_stats.txt
