# generic same-lease A/B of environment settings: [STEPS=n] [BENCH_ARGS="--size 384 576"] bash tools/run_ab_env.sh OUTFILE ROUNDS "ENV1=a ENV2=b" "ENV3=c" ...   ("-" = defaults)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$1; rounds=$2; shift 2
mkdir -p $(dirname $out)
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data $BENCH_ARGS"
for r in $(seq 1 $rounds); do for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e="CDETR_NOTHING=1"
  echo -n "round $r [$v]: "; env $e timeout 300 python bench.py --mode graph --steps ${STEPS:-30} --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done; done | tee $out
