cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4m
mkdir -p $O
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2; do
  for s in "CDETR_PF_POST_US=0 CDETR_TAIL_INLINE=0" "CDETR_PF_POST_US=30 CDETR_TAIL_INLINE=0" "CDETR_PF_POST_US=60 CDETR_TAIL_INLINE=0" "CDETR_PF_POST_US=30 CDETR_TAIL_INLINE=0.5" "CDETR_PF_POST_US=30 CDETR_TAIL_INLINE=1.0" "CDETR_PF_POST_US=30 CDETR_TAIL_INLINE=0.3"; do ab "$s"; done
done 2>&1 | tee $O/ab_post_tail.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
cp $f $O/kernel_trace.csv
python tools/step_phases.py $f $O/step_phases.txt 16
