cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 8 --warmup 3 --no-cpu-baseline --no-alt --no-extra > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-200
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_gap.py $f | tail -10
