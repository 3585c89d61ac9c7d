# round 5, lease A: new timed-path tests, baseline bench (+ crowded shapes), GPU_MAX_HW_QUEUES A/B, crowded-step LSAP time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests/test_timed_path_gpu.py -x -q -s > $O/tests_timed.log 2>&1; echo "timed-path tests rc=$?"
tail -5 $O/tests_timed.log
F="--no-cpu-baseline --no-alt --no-inference --no-real-data"
python bench.py $F > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-200
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5a/bench.log").read().strip().splitlines()[-1])
print("streams", r.get("streams"))
for x in r.get("extra_shapes", []):
    print(x["image"], x["targets"], x["queries"], "%.3f ms" % x["ms_per_step"])
PY
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py $F --no-extra --mode graph > $O/bench_q$q.log 2>&1
  python - <<PY
import json
r = json.loads(open("$O/bench_q$q.log").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES=$q", "%.3f ms" % r["ms_per_step"], r.get("streams"))
PY
done
for T in "37 2100" "3000 3731"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python tools/crowded_step.py $T > $O/crowded_$(echo $T | tr ' ' '_').log 2>&1
  tail -1 $O/crowded_$(echo $T | tr ' ' '_').log
  f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1)
  python tools/kernel_stats.py $f 12 > $O/crowded_$(echo $T | tr ' ' '_')_kernels.txt 2>&1
  grep -i "lsap\|step equiv" $O/crowded_$(echo $T | tr ' ' '_')_kernels.txt
  rm -rf /tmp/prof_c
done
