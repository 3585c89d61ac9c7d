cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_hip_kernels.py -x -q -m gpu -k "replay or encoder_layer or decoder_stack or rcda" > gpurun_out/r2f/tests.log 2>&1
tail -3 gpurun_out/r2f/tests.log
CDETR_BENCH_SHAPES=gpurun_out/r2f/shapes.csv python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2f/bench_shapes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python bench.py --mode graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2f/bench_prof.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2f/bench_kernel_stats.csv \;
python tools/kernel_stats.py gpurun_out/r2f/bench_kernel_stats.csv 60 > gpurun_out/r2f/kernel_summary.txt 2>&1 || true
tail -1 gpurun_out/r2f/bench_prof.log | cut -c1-400
