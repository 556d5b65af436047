#!/bin/bash
# rocprofv3 kernel-trace stats of the replicated-Q step over the peer windows at a world of one (the N > 1 headline's code path:
# bpr_step_fused emitting dQ into the window + peer_allreduce_adam), fused and -- CREC_DP_UNFUSED=1 -- as all-reduce + bpr_adam_dense.
# Output: gpurun_out/profiles/${ROUND:-round3}_dp_peer_kernel_stats.txt
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/profiles
R=${ROUND:-round3}
mkdir -p $OUT
export TMPDIR=/tmp CREC_TRANSPORT=peer CREC_FORCE_DP=1
cd /tmp
: > $OUT/${R}_dp_peer_kernel_stats.txt
for unf in 0 1; do
  export CREC_DP_UNFUSED=$unf
  rm -rf $OUT/trace_dp$unf
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dp$unf -- python $ROOT/bench.py --steps 1824 --warmup 456 --no-extras > $OUT/dp${unf}_under_trace.json 2>/dev/null
  f=$(find $OUT/trace_dp$unf -name "*kernel_stats.csv" | head -1)
  python - "$f" "$unf" >> $OUT/${R}_dp_peer_kernel_stats.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("CREC_TRANSPORT=peer CREC_FORCE_DP=1 CREC_DP_UNFUSED=%s rocprofv3 --kernel-trace --stats -- python bench.py --steps 1824 --warmup 456 --no-extras" % sys.argv[2])
print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows[:8]:
    m = re.search(r"(bpr_\w+|peer_\w+)(<[^>]*>)?", r["Name"])
    name = (m.group(0) if m else r["Name"].split("(")[0])[:60]
    print("%-60s %8s %14s %12.1f %8s" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
print()
PY
  rm -rf $OUT/trace_dp$unf
done
cat $OUT/${R}_dp_peer_kernel_stats.txt
