cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for t in 384 512 768 1024; do
CDETR_WGRAD_TARGET=$t CDETR_BENCH_SHAPES=gpurun_out/shapes_$t.csv python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$t', j['ms_per_step'], j['roofline']['families']['wgrad'])"
done
