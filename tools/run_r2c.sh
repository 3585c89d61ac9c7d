cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
python -X faulthandler tools/bwd_precision.py > gpurun_out/r2c/bwd_precision.txt 2>&1
tail -40 gpurun_out/r2c/bwd_precision.txt
