# Round-6 measurement set: the default bench line, host cost of the launches, rocprofv3 kernel stats of the same step (as it runs: chain layout
# + prefetch; and "unoverlapped": round 3's single layout with every overlap off), phase tables, PMC traffic / SQ passes (own runs, --kernel-trace
# only), per-shape GEMM table.  Outputs under gpurun_out/r6meas/; the summaries are copied to profiles/r6_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6meas
mkdir -p $O
timeout 900 python bench.py > $O/bench_full.log 2> $O/bench_full.err
tail -1 $O/bench_full.log | cut -c1-300
for s in "CDETR_GRAPH_LAYOUT=single" "CDETR_GRAPH_LAYOUT=chain"; do echo "== $s"; env $s python tools/host_cost.py 2>&1 | tail -3; done > $O/host_cost.txt 2>&1
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --mode graph --steps 20 --warmup 5 $F > $O/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python tools/kernel_stats.py $O/bench_kernel_stats.csv 70 > $O/kernel_summary.txt 2>&1
CDETR_GRAPH_LAYOUT=single CDETR_FROZEN_PREFETCH=0 CDETR_WGRAD_EVERY=0 CDETR_BRANCH_BESIDE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k0 -- python bench.py --mode graph --steps 20 --warmup 5 $F > $O/bench_prof0.log 2>&1
find /tmp/prof_k0 -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_unoverlapped.csv \;
python tools/kernel_stats.py $O/bench_kernel_stats_unoverlapped.csv 40 > $O/kernel_summary_unoverlapped.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_phases.py $f $O/step_phases_800x800.txt 16 > /dev/null
python tools/step_phases.py $f $O/step_phases_800x800_inline.txt 3 > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_u -- python bench.py --mode graph --size 384 576 --steps 6 --warmup 2 $F > $O/bench_trace2.log 2>&1
f=$(find /tmp/prof_u -name "*kernel_trace.csv")
python tools/step_phases.py $f $O/step_phases_384x576.txt 16 > /dev/null
B="python bench.py --steps 1 --warmup 1 --no-graph $F"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -- $B > $O/pmc_fetch.log 2>&1
find /tmp/prof_f -name "*counter_collection.csv" -exec cp {} /tmp/fetch.csv \;
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -- $B > $O/pmc_write.log 2>&1
find /tmp/prof_w -name "*counter_collection.csv" -exec cp {} /tmp/write.csv \;
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/prof_s -- $B > $O/pmc_sq.log 2>&1
find /tmp/prof_s -name "*counter_collection.csv" -exec cp {} /tmp/sq.csv \;
python tools/pmc_families.py /tmp/fetch.csv /tmp/write.csv /tmp/sq.csv $O/traffic.json > $O/traffic.txt 2>&1
python tools/pmc_traffic.py /tmp/fetch.csv /tmp/write.csv > $O/pmc_traffic.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv igemm_fast > $O/pmc_sq_igemm_fast.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv igemm_dl > $O/pmc_sq_igemm_dl.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv wgrad_tr > $O/pmc_sq_wgrad_tr.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv rcda > $O/pmc_sq_rcda.txt 2>&1
CDETR_BENCH_SHAPES=$O/shapes.csv python bench.py $F > /dev/null 2>&1
ls -la $O
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_listing.py $f "" adamw_finish 15 > $O/step_listing_pipelined.txt 2>&1
