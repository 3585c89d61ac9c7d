set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
python tools/torch_kernels.py > gpurun_out/r2b/torch_kernels.txt 2>&1
CDETR_BENCH_SHAPES=gpurun_out/r2b/shapes.csv python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2b/bench_shapes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2b/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2b/bench_kernel_stats.csv \;
find /tmp/prof_k -name "*kernel_trace.csv" -exec cp {} /tmp/kt.csv \;
python tools/kernel_stats.py gpurun_out/r2b/bench_kernel_stats.csv > gpurun_out/r2b/kernel_summary.txt 2>&1 || true
head -50 gpurun_out/r2b/torch_kernels.txt
