# round 5: workgroups of the step's LAST weight-gradient launch (layer2's, in line on the main stream, alone on the chip)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5split
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for r in 1 2; do for t in 0 4096 6144 8192 12288 16384; do
  echo -n "round $r tail wgrad target $t: "; CDETR_TAIL_WG_TARGET=$t timeout 300 python bench.py --mode graph --steps 30 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done; done | tee $O/ab_tail_wgrad2.txt
