"""Phases of ONE captured train step from a rocprofv3 --kernel-trace CSV: kernel time and idle gaps between the phase markers
(stem_pack -> gn_fwd -> first flash::fwd -> match_cost -> criterion_bwd -> first rcda_bwd<2,5 (encoder) -> gn_bwd -> sumsq -> adamw).

    python tools/step_phases.py kernel_trace.csv [out.txt]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last complete step: from the last stem_pack that is followed by an adamw_finish
starts = [i for i, r in enumerate(rows) if "stem_pack" in r[2]]
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
s0 = e0 = None
for s in reversed(starts):
    e = next((x for x in ends if x > s), None)
    if e is not None:
        s0, e0 = s, e
        break
# the gradient arena's zero-fill / weight images precede stem_pack in a captured step: walk back to the previous adamw_finish
prev_end = max([x for x in ends if x < s0], default=-1)
step = rows[prev_end + 1:e0 + 1]
markers = [("backbone fwd", "stem_pack"), ("proj + encoder fwd", "gn_fwd"), ("decoder fwd + heads", "flash::fwd"), ("matcher + criterion", "match_cost"),
           ("heads + decoder bwd", "criterion_bwd"), ("encoder bwd", "rcda_bwd_kernel<2, 5"), ("proj + backbone bwd", "gn_bwd"), ("clip + AdamW", "sumsq")]
idx = []
for name, key in markers:
    i = next((k for k, r in enumerate(step) if key in r[2] and (not idx or k > idx[-1][1])), None)
    if i is not None:
        idx.append((name, i))
out = []
t0, t1 = step[0][0], step[-1][1]
out.append("one captured step: %d launches, wall %.3f ms, kernel time %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in step) / 1e6))
pre = step[:idx[0][1]]
if pre:
    out.append("  %-26s launches %4d  kernel %.3f ms  wall %.3f ms" % ("(pre: zero-fill / images)", len(pre), sum(r[1] - r[0] for r in pre) / 1e6, (step[idx[0][1]][0] - t0) / 1e6))
for j, (name, i) in enumerate(idx):
    hi = idx[j + 1][1] if j + 1 < len(idx) else len(step)
    seg = step[i:hi]
    wall = ((step[hi][0] if hi < len(step) else t1) - seg[0][0]) / 1e6
    kt = sum(r[1] - r[0] for r in seg) / 1e6
    top = {}
    for r in seg:
        k = r[2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]
        top[k] = top.get(k, 0) + (r[1] - r[0]) / 1e6
    tops = ", ".join("%s %.2f" % (k, v) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:4])
    out.append("  %-26s launches %4d  kernel %.3f ms  wall %.3f ms   | %s" % (name, len(seg), kt, wall, tops))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
