"""Timeline of ONE chunk of large batches from a rocprofv3 --kernel-trace csv: every launch between two bpr_reduce_loss kernels --
name, start offset, duration, idle gap before it -- with the step launches folded into one line.  usage: trace_chunk_timeline.py DIR"""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"rocprim::(ROCPRIM_[0-9_]+NS::)?detail::", "rocprim::", n)
    return re.sub(r"\(.*", "", n)[:60]
ends = [k for k, r in enumerate(rows) if "bpr_reduce_loss" in r[2]]
a, b = ends[-2], ends[-1]  # the last chunk
t0 = rows[a][1]
prev_end = t0
steps, step_ns, step_gap = 0, 0, 0
busy = 0
print("%-62s %10s %10s %8s" % ("kernel", "start_us", "dur_us", "gap_us"))
for s, e, n in rows[a + 1:b + 1]:
    gap = s - prev_end
    busy += e - s
    if "bpr_triples_lazy" in n:
        steps += 1; step_ns += e - s; step_gap += max(gap, 0)
    else:
        print("%-62s %10.1f %10.1f %8.1f" % (short(n), (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3))
    prev_end = max(prev_end, e)
print("bpr_triples_lazy x %d: %.1f us each, %.1f us of gaps in front of them in total" % (steps, step_ns / max(steps, 1) / 1e3, step_gap / 1e3))
print("chunk span %.1f us, kernels busy %.1f us" % ((rows[b][1] - t0) / 1e3, busy / 1e3))
