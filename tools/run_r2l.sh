cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lvis or rehearsal" > gpurun_out/r2l/tests.log 2>&1
tail -12 gpurun_out/r2l/tests.log
