"""From a rocprofv3 --kernel-trace CSV: for the last captured step, how much of the weight-gradient kernels' time overlaps other kernels.
    python tools/overlap_check.py kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0, t1 = step[0][0], step[-1][1]
wg = [r for r in step if "wgrad" in r[2]]
oth = [r for r in step if "wgrad" not in r[2]]
tot = sum(e - s for s, e, _ in wg)
ov = 0
for s, e, _ in wg:
    for s2, e2, _ in oth:
        lo, hi = max(s, s2), min(e, e2)
        if hi > lo:
            ov += hi - lo
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, kernel time {sum(e - s for s, e, _ in step) / 1e6:.3f} ms, launches {len(step)}")
print(f"wgrad kernels: {len(wg)} launches, {tot / 1e6:.3f} ms, of which {ov / 1e6:.3f} ms overlap other kernels")
for s, e, n in wg[:40]:
    print(f"   {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:70]}")
