cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data --mode graph"
for rep in 1 2; do
for t in 768 256 384 512 1024 1536; do
  CDETR_WGRAD_TARGET=$t CDETR_WGRAD_GROUP_TARGET=$t python bench.py $F > $O/bench_t$t.log 2>&1
  python - <<PY
import json
r = json.loads(open("$O/bench_t$t.log").read().strip().splitlines()[-1])
print("wgrad workgroup target $t", "%.3f ms" % r["ms_per_step"], "median %.3f" % r["step_ms"]["median"])
PY
done
done
