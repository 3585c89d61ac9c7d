"""Phases of ONE captured train step from a rocprofv3 --kernel-trace CSV: kernel time and idle gaps between the phase markers
(stem_pack -> gn_fwd -> first flash::fwd -> match_cost -> criterion_bwd -> first rcda_bwd<2,5 or rcda_bwd_all<2,5 (encoder) -> gn_bwd -> sumsq -> adamw).

    python tools/step_phases.py kernel_trace.csv [out.txt]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# a step = the kernels between two adamw_finish launches; which one: argv[3] counts from the end (default 2: the last but one full step --
# bench.py's timed (pipelined) steps come before its in-line comparison steps, pass a larger number to look at those)
ends = [i for i, r in enumerate(rows) if "adamw_finish" in r[2]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
e0 = ends[-back]
prev_end = ends[-back - 1]
allq = rows[prev_end + 1:e0 + 1]
# round 4 ("chain" layout): the step's main chain is one hardware queue; side work (zero-fill + weight images, the NEXT batch's frozen
# stage, the weight gradients) runs on other queues beside it -- phases are cut on the main queue, the side queues are summarised
counts = {}
for r in allq:
    counts[r[3]] = counts.get(r[3], 0) + 1
mainq = max(counts, key=counts.get)
step = [r for r in allq if r[3] == mainq]
side = {}
for r in allq:
    if r[3] != mainq:
        side.setdefault(r[3], []).append(r)
first = "stem_pack" if any("stem_pack" in r[2] for r in step) else step[0][2][:40]
# (GroupNorm: gn_fwd_kernel / gn_bwd_kernel until round 5, gn_split_stats_kernel / gn_split_bwd_stats_kernel since)
markers = [("backbone fwd", (first,)), ("proj + encoder fwd", ("gn_fwd", "gn_split_stats")), ("decoder fwd + heads", ("flash::fwd",)), ("matcher + criterion", ("match_cost",)),
           ("heads + decoder bwd", ("criterion_bwd",)), ("encoder bwd", ("rcda_bwd_kernel<2, 5", "rcda_bwd_all_kernel<2, 5")), ("proj + backbone bwd", ("gn_bwd", "gn_split_bwd_stats")),
           ("clip + AdamW", ("sumsq",))]
idx = []
for name, keys in markers:
    i = next((k for k, r in enumerate(step) if any(key in r[2] for key in keys) and (not idx or k > idx[-1][1])), None)
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
t00 = allq[0][0]
for q, v in sorted(side.items()):
    names = {}
    for r in v:
        k = r[2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:36]
        names[k] = names.get(k, 0) + (r[1] - r[0]) / 1e6
    tops = ", ".join("%s %.2f" % (k, x) for k, x in sorted(names.items(), key=lambda kv: -kv[1])[:4])
    out.append("  side queue %-15s launches %4d  kernel %.3f ms  from %.3f to %.3f ms of the step | %s" %
               (q, len(v), sum(r[1] - r[0] for r in v) / 1e6, (v[0][0] - t00) / 1e6, (v[-1][1] - t00) / 1e6, tops))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
