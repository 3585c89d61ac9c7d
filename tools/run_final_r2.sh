cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
export TMPDIR=/tmp
bash tools/run_meas_r2.sh > /dev/null 2>&1
O=gpurun_out/r2m
rm -f $O/fetch.csv $O/write.csv $O/sq.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-extra > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_phases.py $f $O/step_phases_800x800.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_u -- python bench.py --mode graph --size 384 576 --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-extra > $O/bench_trace2.log 2>&1
f=$(find /tmp/prof_u -name "*kernel_trace.csv")
python tools/step_phases.py $f $O/step_phases_384x576.txt > /dev/null
CDETR_BENCH_SHAPES=$O/shapes.csv python bench.py --no-cpu-baseline --no-alt --no-extra > /dev/null 2>&1
python tools/bwd_precision.py > $O/bwd_precision.txt 2>&1
tail -1 $O/bench_full.log | cut -c1-300
