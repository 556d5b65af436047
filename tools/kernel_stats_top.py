import csv, re, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2])]:
    print("%-70s %8s %12.1f" % (re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70], r["Calls"], float(r["AverageNs"])))
