cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 1 0; do
CDETR_FUSED_CRITERION=$c rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$c -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > /dev/null 2>&1
f=$(find /tmp/tr$c -name "*kernel_trace.csv"); echo "fused criterion = $c"; python tools/trace_gaps.py $f
done
