"""The last N launches of a rocprofv3 --kernel-trace csv: name, duration, idle gap before each, with runs of step launches folded into
one line (what the fixed costs of a short call look like on the device).  usage: trace_tail.py DIR [N]"""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
STEP = re.compile(r"bpr_step_fused|peer_allreduce_adam|bpr_triples_lazy|bpr_adam_dense|shard_serve|shard_exchange|shard_owner")
def short(k):
    k = re.sub(r"^void ", "", k.replace("(anonymous namespace)::", ""))
    return re.sub(r"\(.*", "", k)[:50]
rows = rows[-n:]
t0, prev = rows[0][0], rows[0][0]
print("%-52s %10s %9s %8s" % ("kernel", "start_us", "dur_us", "gap_us"))
run = None  # [count, first start, busy, gaps]
def flush():
    global run
    if run:
        print("%-52s %10.1f %9.1f %8.1f   (%d step launches: span %.1f us = %.2f us per launch, busy %.1f, idle between them %.1f)" % (
            "  steps", (run[1] - t0) / 1e3, (run[4] - run[1]) / 1e3, run[5] / 1e3, run[0], (run[4] - run[1]) / 1e3, (run[4] - run[1]) / 1e3 / run[0], run[2] / 1e3, run[3] / 1e3))
    run = None
for s, e, k in rows:
    if STEP.search(k):
        if run is None:
            run = [0, s, 0, 0, e, s - prev]
        else:
            run[3] += max(s - prev, 0)
        run[0] += 1; run[2] += e - s; run[4] = e
    else:
        flush()
        print("%-52s %10.1f %9.1f %8.1f" % (short(k), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = max(prev, e)
flush()
