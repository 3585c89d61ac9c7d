cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
for lib in "" tools/exp/libcdetr_half.so; do
  CDETR_LIB=$lib CDETR_BENCH_SHAPES=gpurun_out/r2h/shapes_$(basename "$lib" .so).csv python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2h/bench_$(basename "$lib" .so).log 2>&1
  tail -1 gpurun_out/r2h/bench_$(basename "$lib" .so).log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('lib [$lib]', 'ms/step %.3f'%d['ms_per_step'], 'wgrad ms %.3f'%f['wgrad']['ms_per_step'], 'igemm ms %.3f'%f['igemm']['ms_per_step'])"
done
