cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python -m pytest tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -1; done
python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.log; echo; head -c 300 gpurun_out/bench_default.log
