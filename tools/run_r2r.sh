cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2r
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_hip_kernels.py -x -q -m gpu -s -k "trajectory or layer_norm or adamw or wgrad_group or sumsq" > gpurun_out/r2r/tests.log 2>&1
grep -E "oracle loss|hip    loss|passed|failed|Error" gpurun_out/r2r/tests.log | head
