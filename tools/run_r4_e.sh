# Round 4, run 5: prefetch released behind the cost-matrix graph -- tests, A/B, kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
python -m pytest tests/test_graph_cache.py tests/test_gemm_dl.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_FROZEN_PREFETCH=0" "CDETR_FROZEN_PREFETCH=1"; do ab $s; done
done 2>&1 | tee $O/ab_prefetch.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
cp $f $O/kernel_trace.csv
python -m pytest tests/test_model_gpu.py tests/test_dp_shared_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; tail -3 $O/tests2.log
