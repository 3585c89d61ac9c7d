cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_hip_kernels.py -x -q -k "mirror or sumsq" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-alt > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-160
f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1)
grep -i "weight_mirror\|sumsq\|adamw_kernel\|maxpool" $f | cut -d, -f1-7 | cut -c1-200
