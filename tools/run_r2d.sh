cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
timeout 2400 python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py -x -q -m gpu -k "reduced_term or test_model_gpu" > gpurun_out/r2d/tests_rest.log 2>&1
tail -15 gpurun_out/r2d/tests_rest.log
python bench.py --no-cpu-baseline --no-extra > gpurun_out/r2d/bench.log 2>&1
tail -1 gpurun_out/r2d/bench.log | cut -c1-900
