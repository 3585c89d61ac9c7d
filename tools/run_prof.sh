cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r1m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r1m/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1m/bench_kernel_stats.csv \;
tail -1 gpurun_out/r1m/bench_prof.log | cut -c1-200
