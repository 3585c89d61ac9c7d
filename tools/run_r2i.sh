cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2i/tests_all.log 2>&1
tail -5 gpurun_out/r2i/tests_all.log
bash tools/run_meas_r2.sh
