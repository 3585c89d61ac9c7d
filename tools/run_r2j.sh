cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad or conv or linear or reduced" > gpurun_out/r2j/tests.log 2>&1
tail -3 gpurun_out/r2j/tests.log
for x in 0 1; do
  CDETR_WGRAD_XCD=$x python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2j/bench_x$x.log 2>&1
  tail -1 gpurun_out/r2j/bench_x$x.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('xcd $x', 'ms/step %.3f'%d['ms_per_step'], 'wgrad ms %.3f TF %.0f'%(f['wgrad']['ms_per_step'], f['wgrad']['tflops']))"
done
