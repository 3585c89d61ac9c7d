"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [name_substring ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
counts 128-B read requests at 64 B, so it is doubled; WRITE_SIZE is uncalibrated and taken as reported.
Prints one line per kernel family: launches, corrected read / write bytes per launch.
"""
import csv
import sys
from collections import defaultdict


def family(name):
    for key in ("igemm_fast", "igemm_direct", "igemm_kernel", "wgrad_tr", "wgrad_fast", "wgrad_direct", "wgrad_kernel", "rcda_fwd", "rcda_bwd",
                "rcda_dv", "mha_fwd", "mha_bwd", "lsap", "match_cost", "adamw", "sumsq", "ln_fwd", "ln_bwd", "maxpool"):
        if key in name:
            return key
    return "other:" + name[:60]


def collect(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    want = sys.argv[3:]
    print("%-28s %9s %16s %16s %16s" % ("family", "launches", "read B/launch", "write B/launch", "total B/launch"))
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if want and not any(w in k for w in want):
            continue
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        rd = 2.0 * vf * 1024.0 / max(nf, 1)          # gfx950 correction: x2
        wr = vw * 1024.0 / max(nw, 1)
        rows.append((rd * nf + wr * nw, k, max(nf, nw), rd, wr))
    for _, k, n, rd, wr in sorted(rows, reverse=True)[:40]:
        print("%-28s %9d %16.0f %16.0f %16.0f" % (k[:28], n, rd, wr, rd + wr))


if __name__ == "__main__":
    main()
