"""Summarise a rocprofv3 kernel_stats CSV: python tools/kernel_stats.py stats.csv [top_n] [steps_equiv]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "anonymous namespace)::" in r["Name"] and "at::native" not in r["Name"])
print("total ms %.2f  (per step %.2f)   hand-written HIP %.1f%%   torch/rocBLAS %.1f%%   launches/step %.0f" % (
    tot / 1e6, tot / 1e6 / steps, 100 * ours / tot, 100 * (1 - ours / tot), sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:n]:
    print("%9.2f ms %6.2f%% calls %6d avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"]), int(r["Calls"]) / steps,
                                                          float(r["AverageNs"]) / 1e3, r["Name"][:120]))
