cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for p in 1024 512 256 64; do
  CDETR_WGRAD_DIRECT_MAXP=$p python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2p/bench_p$p.log 2>&1
  tail -1 gpurun_out/r2p/bench_p$p.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('maxp $p', 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'wgrad ms %.3f'%f['wgrad']['ms_per_step'])"
done
