cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "pixel_blocks or reduced_term or conv_forward_dgrad_wgrad" > $O/tests_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -5 $O/tests_wgrad.log
timeout 600 python tools/wgrad_twin_bench.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | tee $O/wgrad_kp_bench.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data --mode graph"
for kp in 1 2 4 1 2 4; do
  CDETR_WGRAD_KP=$kp python bench.py $F > $O/bench_kp$kp.log 2>&1
  python - <<PY
import json
r = json.loads(open("$O/bench_kp$kp.log").read().strip().splitlines()[-1])
print("CDETR_WGRAD_KP=$kp", "%.3f ms" % r["ms_per_step"], "median %.3f" % r["step_ms"]["median"])
PY
done
