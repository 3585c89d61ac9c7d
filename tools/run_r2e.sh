cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2e/tests_all.log 2>&1
tail -25 gpurun_out/r2e/tests_all.log
python bench.py --no-cpu-baseline --no-extra > gpurun_out/r2e/bench.log 2>&1
tail -1 gpurun_out/r2e/bench.log | cut -c1-700
python tools/torch_kernels.py > gpurun_out/r2e/torch_kernels.txt 2>&1
head -32 gpurun_out/r2e/torch_kernels.txt | cut -c1-150
