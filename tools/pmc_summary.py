"""Per-kernel averages of rocprofv3 --pmc counters: python tools/pmc_summary.py counter_collection.csv [name filter]"""
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if filt and filt not in k:
        continue
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("    %-28s %16.0f" % (c, agg[k][c] / cnt[k][c]))
