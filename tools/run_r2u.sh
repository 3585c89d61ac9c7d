cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2u
timeout 1200 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py tests/test_abi.py -x -q -m "gpu or not gpu" -k "bf16x3 or abi or trajectory or replay or dataset" > gpurun_out/r2u/tests.log 2>&1
tail -4 gpurun_out/r2u/tests.log
for t in 0 1; do
  CDETR_TWINS=$t python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2u/b$t.log 2>&1
  tail -1 gpurun_out/r2u/b$t.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('twins $t', 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'wgrad ms %.3f'%f['wgrad']['ms_per_step'], 'igemm ms %.3f'%f['igemm']['ms_per_step'])"
done
