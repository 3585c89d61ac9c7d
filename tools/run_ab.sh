# A/B of one environment knob on the bench step, alternating runs on the same box.  usage: bash tools/run_ab.sh "A=0" "A=1" [repeats]
cd $GRAFT_REPO_ROOT
n=${3:-2}
for i in $(seq 1 $n); do
  for s in "$1" "$2"; do
    env $s python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$s', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
  done
done
