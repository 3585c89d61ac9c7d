cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q
mkdir -p $O
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_LSAP_PRIO=0" "CDETR_LSAP_PRIO=1"; do ab "$s"; done
done 2>&1 | tee $O/ab_prio.txt
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "lsap or matcher or criterion" > $O/tests.log 2>&1; tail -2 $O/tests.log
