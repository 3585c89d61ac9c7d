cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2k
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2k/tests_all.log 2>&1
tail -5 gpurun_out/r2k/tests_all.log
python bench.py --no-cpu-baseline --no-extra --no-alt > gpurun_out/r2k/bench.log 2>&1
tail -1 gpurun_out/r2k/bench.log | cut -c1-300
