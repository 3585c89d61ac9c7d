cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p
mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee $O/prio.txt
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_SIDE_PRIORITY=normal" "CDETR_SIDE_PRIORITY=low"; do ab "$s"; done
done 2>&1 | tee $O/ab_priority.txt
python -m pytest tests/test_dp_shared_gpu.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
