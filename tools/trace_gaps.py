"""Idle-gap analysis of graph-replay steps from a rocprofv3 kernel trace: python tools/trace_gaps.py kernel_trace.csv"""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2] and "finish" not in r[2]]
print("kernels", len(rows), "adamw launches", len(ends))
# the last 5 steps
sel = ends[-6:]
for a, b in zip(sel[:-1], sel[1:]):
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    print("step: %d kernels  span %.2f ms  busy %.2f ms  idle %.2f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
seg = rows[sel[-2] + 1:sel[-1] + 1]
gaps = defaultdict(lambda: [0, 0.0])
big = []
for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
    g = s1 - e0
    key = (n0[:50], n1[:50])
    gaps[key][0] += 1; gaps[key][1] += g
    big.append((g, n0[:60], n1[:60]))
tot = sum(g for g, _, _ in big)
print("sum of gaps %.2f ms over %d transitions (avg %.2f us)" % (tot / 1e6, len(big), tot / len(big) / 1e3))
import statistics
print("median gap %.2f us" % (statistics.median(g for g, _, _ in big) / 1e3))
for g, a, b in sorted(big, reverse=True)[:12]:
    print("%8.1f us  %s -> %s" % (g / 1e3, a, b))
