cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for g in 0 1; do
  CDETR_GROUPS=$g rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t$g -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/trace_g$g.log 2>&1
  f=$(find /tmp/prof_t$g -name "*kernel_trace.csv")
  python tools/step_phases.py $f $O/step_phases_groups$g.txt 16 > /dev/null
  echo "== CDETR_GROUPS=$g"; head -4 $O/step_phases_groups$g.txt | cut -c1-110
  CDETR_GROUPS=$g CDETR_BENCH_SHAPES=$O/shapes_g$g.csv python bench.py $F > /dev/null 2>&1
done
