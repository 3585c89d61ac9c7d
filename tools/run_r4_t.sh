# Round 4, run 20: launch-count cuts (S12 piece merge, merged encoder prologue, fused positional-MLP node)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
python -m pytest tests/test_hip_kernels.py tests/test_graph_cache.py tests/test_model_gpu.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_X=0" "CDETR_S_PIECES=3" "CDETR_FUSED_HEADS=0"; do ab "$s"; done
done 2>&1 | tee $O/ab_launches.txt
