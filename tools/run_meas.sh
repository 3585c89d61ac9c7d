set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1v
python bench.py > gpurun_out/r1v/bench_full.log 2>&1
tail -1 gpurun_out/r1v/bench_full.log | cut -c1-1500
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r1v/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1v/bench_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-alt > gpurun_out/r1v/pmc_fetch.log 2>&1
find /tmp/prof_f -name "*counter_collection.csv" -exec cp {} gpurun_out/r1v/fetch_counter_collection.csv \;
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-alt > gpurun_out/r1v/pmc_write.log 2>&1
find /tmp/prof_w -name "*counter_collection.csv" -exec cp {} gpurun_out/r1v/write_counter_collection.csv \;
ls -la gpurun_out/r1v
head -3 gpurun_out/r1v/fetch_counter_collection.csv
python tools/pmc_traffic.py gpurun_out/r1v/fetch_counter_collection.csv gpurun_out/r1v/write_counter_collection.csv > gpurun_out/r1v/pmc_traffic.txt 2>&1
cat gpurun_out/r1v/pmc_traffic.txt | head -30
