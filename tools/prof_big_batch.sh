#!/bin/bash
# rocprofv3 kernel-trace stats of the S-big step at a large batch (index_big.hip's phases beside bpr_triples_lazy).
# usage (on the GPU box): tools/prof_big_batch.sh BATCH STEPS OUTNAME
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
B=${1:-65536}; S=${2:-64}; NAME=${3:-bigbatch}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python $ROOT/tools/bench_big.py --batch $B --steps $S > $OUT/${NAME}_under_trace.json 2>/dev/null
f=$(find $OUT/trace_$NAME -name "*kernel_stats.csv" | head -1)
python - "$f" "tools/bench_big.py --batch $B --steps $S" > $OUT/${NAME}_kernel_stats.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python " + sys.argv[2])
print("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows[:30]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"rocprim::(ROCPRIM_[0-9_]+NS::)?detail::", "rocprim::", name)[:70]
    print("%-70s %8s %14s %12.1f %8s" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
PY
rm -rf $OUT/trace_$NAME
cat $OUT/${NAME}_kernel_stats.txt
