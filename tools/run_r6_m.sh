cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --mode graph --steps 20 --warmup 5 $F > $O/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python tools/kernel_stats.py $O/bench_kernel_stats.csv 70 > $O/kernel_summary.txt 2>&1
grep -i "direct\|ln_fwd\|ln_bwd\|rcda" $O/kernel_summary.txt | cut -c1-170
