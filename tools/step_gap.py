"""Idle time between consecutive captured steps (graph replays) from a rocprofv3 --kernel-trace CSV.
    python tools/step_gap.py kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
for a, b in zip(ends[:-1], ends[1:]):
    step = rows[a + 1:b + 1]
    gap = (step[0][0] - rows[a][1]) / 1e3
    wall = (step[-1][1] - step[0][0]) / 1e3
    # idle inside the step: time not covered by any kernel
    cover, cur = 0, step[0][0]
    for s, e, _ in step:
        if e > cur:
            cover += e - max(s, cur)
            cur = e
    print(f"gap before step {gap:8.1f} us   step wall {wall:9.1f} us   idle inside {(wall * 1e3 - cover) / 1e3:7.1f} us  launches {len(step)}  first kernel {step[0][2][:50]}")
