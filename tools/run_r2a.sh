set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py tests/test_dp_shared_gpu.py -x -q -m gpu -s > gpurun_out/r2a/tests_new.log 2>&1
tail -30 gpurun_out/r2a/tests_new.log
timeout 600 python bench.py > gpurun_out/r2a/bench.log 2>&1
tail -1 gpurun_out/r2a/bench.log | cut -c1-3000
