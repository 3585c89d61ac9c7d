"""Kernel-by-kernel listing of the last captured step between two marker kernels, from a rocprofv3 --kernel-trace CSV.
    python tools/step_listing.py kernel_trace.csv START_MARKER END_MARKER [STEPS_BACK]
STEPS_BACK = 0 (default): the last step of the run (a pipelined replay prefetches nothing there); 1: the step before it (steady state)."""
import csv, sys, re
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
back = int(sys.argv[4]) if len(sys.argv) > 4 else 0
step = rows[ends[-2 - back] + 1:ends[-1 - back] + 1]
a = next(i for i, r in enumerate(step) if sys.argv[2] in r[2])
b = next(i for i, r in enumerate(step) if sys.argv[3] in r[2] and i > a)
prev = step[a][0]
for s, e, n, g, w in step[a:b]:
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    print(f"{(s - step[a][0]) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} gap {(s - prev) / 1e3:5.1f}  grid {g:>8s}/{w:<4s} {n[:90]}")
    prev = e
