cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2o
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2o/tests_all.log 2>&1
tail -4 gpurun_out/r2o/tests_all.log
python bench.py --no-cpu-baseline --no-extra --no-alt > gpurun_out/r2o/bench.log 2>&1
tail -1 gpurun_out/r2o/bench.log | cut -c1-300
python tools/torch_kernels.py > gpurun_out/r2o/torch_kernels.txt 2>&1
head -12 gpurun_out/r2o/torch_kernels.txt | tail -10
