cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5split
mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for r in 1 2; do for v in "CDETR_TAIL_INLINE=1.0" "CDETR_TAIL_INLINE=0.85" "CDETR_TAIL_INLINE=0.7" "CDETR_TAIL_INLINE=0.5" "CDETR_WGRAD_GROUP_TARGET=512" "CDETR_WGRAD_GROUP_TARGET=768" "CDETR_WGRAD_TARGET=512" "CDETR_S_PIECES=2"; do
  echo -n "round $r $v: "; env $v timeout 300 python bench.py --mode graph --steps 30 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done; done | tee $O/ab_tail_retune.txt
