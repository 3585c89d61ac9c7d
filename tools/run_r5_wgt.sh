cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5split
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for r in 1 2; do for t in 256 384 512; do for g in 128 192 256 384 512; do
  echo -n "round $r wgrad target $t group target $g: "; CDETR_WGRAD_TARGET=$t CDETR_WGRAD_GROUP_TARGET=$g timeout 300 python bench.py --mode graph --steps 30 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done; done; done | tee $O/ab_wgrad_targets.txt
