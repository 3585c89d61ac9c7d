# Round 4, run 4: CU-masked prefetch stream A/B; interleaved-groups forward sweep; trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
python -m pytest tests/test_graph_cache.py -m gpu -x -q -k "prefetch or pipelined" > $O/tests.log 2>&1; tail -3 $O/tests.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2; do
  for s in "CDETR_FROZEN_PREFETCH=0" "CDETR_PF_FREE_CUS=0" "CDETR_PF_FREE_CUS=16" "CDETR_PF_FREE_CUS=32" "CDETR_PF_FREE_CUS=64"; do ab $s; done
done 2>&1 | tee $O/ab_cumask.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 3 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
cp $f $O/kernel_trace.csv
DL_SWEEP_GROUPS=1 python tools/dl_sweep.py fwd > $O/fwd_split_groups.txt 2>&1
cat $O/fwd_split_groups.txt | cut -c1-200
