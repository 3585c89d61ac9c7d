# Round-2 measurement set (called by tools/run_final_r2.sh, which adds the phase traces, the per-shape table and the backward-precision table): the default bench line, rocprofv3 kernel stats of the same command, PMC traffic / SQ passes (own runs,
# --kernel-trace only).  Outputs under gpurun_out/r2m/; the summaries are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
python bench.py > $O/bench_full.log 2>&1
tail -1 $O/bench_full.log | cut -c1-1200
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-extra > $O/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python tools/kernel_stats.py $O/bench_kernel_stats.csv 70 > $O/kernel_summary.txt 2>&1
B="python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-alt --no-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -- $B > $O/pmc_fetch.log 2>&1
find /tmp/prof_f -name "*counter_collection.csv" -exec cp {} /tmp/fetch.csv \;
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -- $B > $O/pmc_write.log 2>&1
find /tmp/prof_w -name "*counter_collection.csv" -exec cp {} /tmp/write.csv \;
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/prof_s -- $B > $O/pmc_sq.log 2>&1
find /tmp/prof_s -name "*counter_collection.csv" -exec cp {} /tmp/sq.csv \;
cp /tmp/fetch.csv /tmp/write.csv /tmp/sq.csv $O/ 2>/dev/null; python tools/pmc_families.py /tmp/fetch.csv /tmp/write.csv /tmp/sq.csv $O/traffic.json > $O/traffic.txt 2>&1
python tools/pmc_traffic.py /tmp/fetch.csv /tmp/write.csv > $O/pmc_traffic.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv igemm_fast > $O/pmc_sq_igemm_fast.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv wgrad_tr > $O/pmc_sq_wgrad_tr.txt 2>&1
python tools/pmc_summary.py /tmp/sq.csv rcda > $O/pmc_sq_rcda.txt 2>&1
ls -la $O
head -40 $O/traffic.txt
