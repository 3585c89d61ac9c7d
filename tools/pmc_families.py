"""Per-family HBM traffic and SQ-pipe occupancy of one eager bench step from rocprofv3 --pmc passes -> profiles/r3_traffic.json.

    python tools/pmc_families.py fetch.csv write.csv sq1.csv out.json

FETCH_SIZE / WRITE_SIZE come from separate passes (they cannot share one on gfx950) and are reported in KiB; per
MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts 128-B read requests at 64 B on gfx950, so it is doubled; WRITE_SIZE is taken as
reported (uncalibrated).  Families follow bench.py's `roofline.families`: a grouped launch counts as ONE launch, like there.
SQ counters (quad-cycles, except SQ_VALU_MFMA_BUSY_CYCLES in cycles): MFMA-pipe busy = MFMA_BUSY / (4 x SQ_BUSY_CYCLES per SE ...) is
reported as the plain ratio SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES) = share of a resident wave's lifetime in which its SIMD's
matrix pipe was busy; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = (("igemm_fewrow", ("igemm_direct", "false, 2>(", "true, 2>(")),      # few-row class: the one-wave-per-tile kernel and the two-wave-group tile
            ("igemm", ("igemm_fast", "igemm_kernel", "igemm_mixed", "igemm_dl")), ("wgrad", ("wgrad_tr", "wgrad_fast", "wgrad_direct", "wgrad_kernel")),
            ("rcda_fwd", ("rcda_fwd",)), ("rcda_bwd", ("rcda_bwd", "rcda_dv")), ("mha", ("flash::",)), ("lsap", ("lsap_",)),
            ("layernorm", ("ln_fwd", "ln_bwd")), ("optimizer", ("adamw", "sumsq")), ("weight_mirror", ("weight_mirror",)))


def fam(name):
    for f, keys in FAMILIES:
        if any(k in name for k in keys):
            return f
    return "other"


def collect(path, counters):
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    with open(path) as f:
        for r in csv.DictReader(f):
            c = r["Counter_Name"]
            if c in counters:
                k = fam(r["Kernel_Name"])
                agg[k][c] += float(r["Counter_Value"])
                n[k][c] += 1
    return agg, n


def main():
    fetch, nf = collect(sys.argv[1], {"FETCH_SIZE"})
    write, nw = collect(sys.argv[2], {"WRITE_SIZE"})
    sq, nsq = collect(sys.argv[3], {"SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                    "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"})
    fams, per_launch, pmc, detail = {}, {}, {}, {}
    # the profiled process runs several step equivalents (warm-up + timed step + bench.py's roofline leg): one criterion launch each;
    # the optimizer kernels run once per FULL step only (one adamw_finish each)
    nstep = nfull = 0
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == "FETCH_SIZE":
                nstep += "criterion_fwd_kernel" in r["Kernel_Name"]
                nfull += "adamw_finish" in r["Kernel_Name"]
    nstep, nfull = max(nstep, 1), max(nfull, 1)
    for k in sorted(set(fetch) | set(write)):
        launches = max(nf[k]["FETCH_SIZE"], nw[k]["WRITE_SIZE"], 1)
        rd = 2.0 * fetch[k]["FETCH_SIZE"] * 1024.0
        wr = write[k]["WRITE_SIZE"] * 1024.0
        div = nfull if k == "optimizer" else nstep          # (the weight images are refreshed by every forward/backward, optimizer step or not)
        fams[k] = (rd + wr) / div
        per_launch[k] = (rd + wr) / launches
        detail[k] = {"kernel_launches_per_step": launches / div, "read_bytes_per_kernel_launch": rd / launches,
                     "write_bytes_per_kernel_launch": wr / launches}
    for k, c in sq.items():
        wc = c["SQ_WAVE_CYCLES"]
        if wc > 0:
            pmc[k] = {"mfma_busy_share_of_wave_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), "wait_any_share": c["SQ_WAIT_ANY"] / wc,
                      "wait_inst_share": c["SQ_WAIT_INST_ANY"] / wc, "active_inst_share": c["SQ_ACTIVE_INST_ANY"] / wc,
                      "valu_share": c["SQ_ACTIVE_INST_VALU"] / wc, "lds_share": c["SQ_ACTIVE_INST_LDS"] / wc}
    out = {"workload": "B=2 800x800 Q=300 bf16x3 fwd / bf16 bwd, one eager step (bench.py --no-graph --steps 1 --warmup 1)",
           "step_equivalents_in_the_profiled_process": nstep, "families_bytes_per_step": fams, "families_bytes_per_kernel_launch": per_launch,
           "detail": detail, "pmc": pmc, "total_bytes_per_step": sum(fams.values()),
           "source": "profiles/r6_pmc_* summaries (tools/run_meas_r6.sh): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* in separate passes with "
                     "--kernel-trace only; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section), KiB -> bytes; "
                     "tools/pmc_families.py"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
