"""Summarise a rocprofv3 kernel_stats CSV: python tools/kernel_stats.py stats.csv [top_n] [steps_equiv]
steps_equiv defaults to the number of criterion launches in the file (= train-step equivalents the profiled process ran: warm-ups,
mode probes, timed steps, the roofline leg), so "per step" figures are per step whatever the command line was."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# (round 4) the one-thread flag_wait / delay kernels only SLEEP until another stream's signal: they occupy no compute unit and are not work
idle = [r for r in rows if "flag_wait_kernel" in r["Name"] or "delay_kernel" in r["Name"]]
rows = [r for r in rows if r not in idle]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
crit = sum(int(r["Calls"]) for r in rows if "criterion_fwd_kernel" in r["Name"])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else float(max(crit, 1))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "anonymous namespace)::" in r["Name"] and "at::native" not in r["Name"])
print("step equivalents %.0f   kernel time total %.2f ms = %.3f ms per step   hand-written HIP %.1f%%   torch %.1f%%   launches per step %.0f" % (
    steps, tot / 1e6, tot / 1e6 / steps, 100 * ours / tot, 100 * (1 - ours / tot), sum(int(r["Calls"]) for r in rows) / steps))
for r in idle:
    print("   (idle: %.3f ms/step in %.1f calls/step of %s -- sleeping, not working)" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, r["Name"][:60]))
for r in rows[:n]:
    print("%9.3f ms/step %6.2f%% calls/step %7.1f avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"]), int(r["Calls"]) / steps,
                                                                     float(r["AverageNs"]) / 1e3, r["Name"][:120]))
