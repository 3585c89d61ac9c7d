cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_dl.py -x -q -k "grouped" > $O/tests_groups.log 2>&1; echo "grouped tests rc=$?"; tail -15 $O/tests_groups.log | cut -c1-300
timeout 1500 python -m pytest tests/test_full_size_gpu.py tests/test_timed_path_gpu.py -x -q > $O/tests_full.log 2>&1; echo "full-size + timed-path tests rc=$?"; tail -5 $O/tests_full.log | cut -c1-300
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data --mode graph"
for rep in 1 2; do
for gset in "0:2,3,4" "1:2,3,4" "1:2,3" "1:3"; do
  on=${gset%%:*}; lay=${gset##*:}
  CDETR_GROUPS=$on CDETR_GROUPS_LAYERS=$lay python bench.py $F > $O/bench_g.log 2>&1
  python - <<PY
import json
r = json.loads(open("$O/bench_g.log").read().strip().splitlines()[-1])
print("CDETR_GROUPS=$on layers $lay", "%.3f ms" % r["ms_per_step"], "median %.3f" % r["step_ms"]["median"], "loss %.5f" % r.get("final_loss", float("nan")))
PY
done
done
