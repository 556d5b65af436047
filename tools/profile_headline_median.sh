#!/bin/bash
# The headline workload's rocprofv3 --kernel-trace --stats summary taken FIVE times on this box; the run with the median bpr_step_fused
# average becomes gpurun_out/profiles/${ROUND:-round3}_kernel_stats.txt, the five averages are listed at its end (the average moves
# by several per cent from box to box and with what ran just before: one run is a draw, not a measurement).
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/profiles
R=${ROUND:-round3}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for k in 1 2 3 4 5; do
  rm -rf /tmp/hm_$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hm_$k -- python $ROOT/bench.py --steps 1824 --warmup 456 --no-extras > /dev/null 2>&1
  f=$(find /tmp/hm_$k -name "*kernel_stats.csv" | head -1)
  python - "$f" > /tmp/hm_$k.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python bench.py --steps 1824 --warmup 456 --no-extras")
print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows:
    m = re.search(r"(bpr_\w+|eval_\w+|rank_metrics\w*)(<[^>]*>)?", r["Name"])
    name = (m.group(0) if m else r["Name"].split("(")[0])[:60]
    print("%-60s %8s %14s %12.1f %8s" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
PY
  python $ROOT/tools/trace_steady.py /tmp/hm_$k bpr_step_fused 456 >> /tmp/hm_$k.txt
done
python - > $OUT/${R}_kernel_stats.txt <<'PY'
import re
runs = []
for k in range(1, 6):
    t = open("/tmp/hm_%d.txt" % k).read()
    avg = float(re.search(r"^bpr_step_fused.*?\s\d+\s+\d+\s+([0-9.]+)\s", t, re.M).group(1))
    runs.append((avg, k, t))
runs.sort()
avg, k, t = runs[2]
print(t.rstrip())
print("(the median of five consecutive runs of this command on one box, tools/profile_headline_median.sh; bpr_step_fused averages of the five, ns: %s)"
      % ", ".join("%.1f" % r[0] for r in sorted(runs, key=lambda r: r[1])))
PY
cat $OUT/${R}_kernel_stats.txt
