cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2w
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2w/tests_all.log 2>&1
tail -4 gpurun_out/r2w/tests_all.log
for t in 0 1; do
  CDETR_GEMM_A16=$t python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2w/b$t.log 2>&1
  tail -1 gpurun_out/r2w/b$t.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('A16 $t', 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'wgrad ms %.3f'%f['wgrad']['ms_per_step'], 'igemm ms %.3f'%f['igemm']['ms_per_step'])"
done
