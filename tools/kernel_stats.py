import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("total ms", round(tot / 1e6, 2))
for r in rows[:n]:
    print("%9.2f ms %6.2f%% calls %6d avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), int(r["Calls"]),
                                                          float(r["AverageNs"]) / 1e3, r["Name"][:110]))
