cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data --mode graph"
for rep in 1 2; do
for v in 0 1 2 3 4; do
  CDETR_TUNING=1 CDETR_WGRAD_VARIANT=$v python bench.py $F > $O/bench_v$v.log 2>&1
  python - <<PY
import json
r = json.loads(open("$O/bench_v$v.log").read().strip().splitlines()[-1])
print("CDETR_WGRAD_VARIANT=$v", "%.3f ms" % r["ms_per_step"], "median %.3f" % r["step_ms"]["median"])
PY
done
done
