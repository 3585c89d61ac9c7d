# Round 4, run 11: chain layout with probed side streams + flag-released prefetch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
python -m pytest tests/test_graph_cache.py -m gpu -x -q -k "graph or prefetch or pipelined or layouts or cached" > $O/tests.log 2>&1; tail -3 $O/tests.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2; do
  for s in "CDETR_GRAPH_LAYOUT=single CDETR_FROZEN_PREFETCH=0" "CDETR_GRAPH_LAYOUT=single" "CDETR_GRAPH_LAYOUT=chain CDETR_FROZEN_PREFETCH=0" "CDETR_GRAPH_LAYOUT=chain" "CDETR_GRAPH_LAYOUT=chain CDETR_PF_TIMEOUT_US=0"; do ab "$s"; done
done 2>&1 | tee $O/ab_layout.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
cp $f $O/kernel_trace.csv
