cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log
python bench.py > $O/bench_full.log 2> $O/bench_full.err; tail -1 $O/bench_full.log | cut -c1-300
