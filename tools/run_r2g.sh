cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
for v in 0 1 2 3 4; do
  CDETR_WGRAD_VARIANT=$v python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2g/bench_wv$v.log 2>&1
  tail -1 gpurun_out/r2g/bench_wv$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('variant $v', 'ms/step %.3f'%d['ms_per_step'], 'wgrad ms %.3f TF %.0f launches %d'%(f['wgrad']['ms_per_step'], f['wgrad']['tflops'], f['wgrad']['launches_per_step']))"
done
