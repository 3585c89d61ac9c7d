"""Idle gaps of the MAIN queue inside one captured step (rocprofv3 --kernel-trace CSV): every pair of consecutive main-queue kernels that
are more than GAP_US apart, with the kernels on either side -- the graph boundaries and dispatch delays of the chain.
    python tools/step_gaps.py kernel_trace.csv [STEPS_BACK=15] [GAP_US=8]"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 15
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
step = rows[ends[-2 - back] + 1:ends[-1 - back] + 1]
cnt = {}
for r in step:
    cnt[r[3]] = cnt.get(r[3], 0) + 1
mainq = max(cnt, key=cnt.get)
main = [r for r in step if r[3] == mainq]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", ""))[:60]
t0 = main[0][0]
tot = 0.0
for a, b in zip(main, main[1:]):
    g = (b[0] - a[1]) / 1e3
    if g > thr:
        tot += g
        side = [r for r in step if r[3] != mainq and r[0] < b[0] and r[1] > a[1]]
        print(f"{(a[1] - t0) / 1e3:8.1f} us: idle {g:6.1f} us between {short(a[2])} and {short(b[2])}" + (f"   [side queue busy: {', '.join(sorted({short(r[2])[:28] for r in side}))[:120]}]" if side else ""))
print(f"main queue: {len(main)} kernels, wall {(main[-1][1] - t0) / 1e3:.1f} us, idle in gaps > {thr:g} us: {tot:.1f} us")
