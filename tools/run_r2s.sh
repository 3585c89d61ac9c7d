cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_abi.py -x -q -k "reduced_term or abi or wgrad" > gpurun_out/r2s/tests.log 2>&1
tail -4 gpurun_out/r2s/tests.log
for e in 0 1; do
  CDETR_TWIN_EXP=$e python bench.py --mode eager --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2s/b$e.log 2>&1
  tail -1 gpurun_out/r2s/b$e.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('twin_exp $e', 'ms/step %.3f'%d['ms_per_step'], 'wgrad ms %.3f TF %.0f'%(f['wgrad']['ms_per_step'], f['wgrad']['tflops']))"
done
