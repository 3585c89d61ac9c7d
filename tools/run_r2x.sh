cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2x
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2x/tests_all.log 2>&1
tail -4 gpurun_out/r2x/tests_all.log
python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2x/b.log 2>&1
tail -1 gpurun_out/r2x/b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'rcda_bwd ms %.3f'%f['rcda_bwd']['ms_per_step'], 'rcda_fwd ms %.3f'%f['rcda_fwd']['ms_per_step'])"
