cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python tools/rcda_bench.py 10 > /dev/null 2>&1
f=$(find /tmp/pr -name "*kernel_stats.csv"); python tools/kernel_stats.py $f 12 1 | cut -c1-150
