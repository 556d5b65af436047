#!/bin/bash
# refresh the models leg of tools/profile.sh only (kernel-trace stats + FETCH/WRITE counters of tools/bench_models.py)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/profiles; R=${ROUND:-round2}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/trace_models $OUT/pmcmodels_*
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_models -- python $ROOT/tools/bench_models.py > $OUT/models_under_trace.json 2>/dev/null
f=$(find $OUT/trace_models -name "*kernel_stats.csv" | head -1)
python - "$f" "tools/bench_models.py" > $OUT/models_kernel_stats.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python " + sys.argv[2])
print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows[:40]:
    m = re.search(r"(bpr_\w+|eval_\w+|rank_metrics\w*|shard_\w+|ncf_\w+|fm_\w+|cml_\w+|rows_\w+|apply_all|small_adam\w*|topk_\w+|lrml_\w+|tcf_\w+|sbpr_\w+|cune_\w+)(<[^>]*>)?", r["Name"])
    name = (m.group(0) if m else r["Name"].split("(")[0])[:60]
    print("%-60s %8s %14s %12.1f %8s" % (name, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
PY
rm -rf $OUT/trace_models/*/*kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmcmodels_$c -- python $ROOT/tools/bench_models.py > /dev/null 2>&1 || echo "pmc models $c failed"
done
cd $ROOT && python tools/pmc_summary.py $OUT/pmcmodels_* > $OUT/models_pmc_summary.txt
cp $OUT/models_kernel_stats.txt $OUT/${R}_models_kernel_stats.txt; cp $OUT/models_pmc_summary.txt $OUT/${R}_models_pmc_summary.txt
cat $OUT/models_kernel_stats.txt
