# round 5, lease C (after the container was re-created): multi-wave LSAP parity, timed-path tests, crowded-step LSAP time, baseline bench + kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "lsap or matcher" > $O/tests_lsap.log 2>&1; echo "lsap tests rc=$?"; tail -3 $O/tests_lsap.log
timeout 1200 python -m pytest tests/test_timed_path_gpu.py -x -q > $O/tests_timed.log 2>&1; echo "timed-path tests rc=$?"; tail -3 $O/tests_timed.log
for T in "37 2100" "3000 3731"; do
  n=$(echo $T | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python tools/crowded_step.py $T > $O/crowded_$n.log 2>&1
  tail -1 $O/crowded_$n.log
  f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1)
  python tools/kernel_stats.py $f 12 > $O/crowded_${n}_kernels.txt 2>&1
  grep -i "lsap\|step equiv" $O/crowded_${n}_kernels.txt
  CDETR_LSAP_GENERIC=1 python tools/crowded_step.py $T 2>&1 | tail -1
  rm -rf /tmp/prof_c
done
python bench.py > $O/bench_full.log 2> $O/bench_full.err; tail -1 $O/bench_full.log | cut -c1-400
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --mode graph --steps 20 --warmup 5 $F > $O/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python tools/kernel_stats.py $O/bench_kernel_stats.csv 70 > $O/kernel_summary.txt 2>&1
head -30 $O/kernel_summary.txt
CDETR_BENCH_SHAPES=$O/shapes.csv python bench.py $F > /dev/null 2>&1
