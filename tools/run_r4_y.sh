# Round 4, run 27: side-stream probe on WARM candidates (a cold stream's first launch looked serialised: the full bench reported 0 of 8 overlapping)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p $O
python -m pytest tests/test_graph_cache.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py > $O/bench_full.log 2> $O/bench_full.err; tail -1 $O/bench_full.log | cut -c1-200; grep -o '"side_streams": {[^}]*}' $O/bench_full.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3), d['frozen_stage_prefetch']['side_streams'])"; }
for i in 1 2 3; do ab "CDETR_X=0"; done 2>&1 | tee $O/ab.txt
