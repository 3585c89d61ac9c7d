cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data --mode graph"
run() {
  env "$@" python bench.py $F > $O/b.log 2>&1
  python - "$*" <<'PY'
import json, sys
try:
    r = json.loads(open("gpurun_out/r5m/b.log").read().strip().splitlines()[-1])
    print("%-44s %.3f ms  median %.3f  streams %s" % (sys.argv[1], r["ms_per_step"], r["step_ms"]["median"], r.get("streams")))
except Exception as e:
    print("%-44s failed: %s" % (sys.argv[1], e))
PY
}
for rep in 1 2 3; do
run X=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=4
done
