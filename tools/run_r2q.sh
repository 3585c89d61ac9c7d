cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
run() { env $1 python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2q/b.log 2>&1; tail -1 gpurun_out/r2q/b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('$1', 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'igemm ms %.3f'%f['igemm']['ms_per_step'])"; }
run X=1
run CDETR_GEMM_F24=1
run CDETR_GEMM_F24=2
run CDETR_GEMM_F24=0
run CDETR_LN_BWD_ROWS=32
run CDETR_LN_BWD_ROWS=8
run X=1
