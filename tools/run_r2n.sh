cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "cfg2 and bf16x3" > gpurun_out/r2n/tests.log 2>&1
tail -3 gpurun_out/r2n/tests.log
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2n/bench.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_phases.py $f gpurun_out/r2n/step_phases.txt
for sz in "384 576"; do
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_u -- python bench.py --mode graph --size $sz --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-extra > gpurun_out/r2n/bench_small.log 2>&1
f=$(find /tmp/prof_u -name "*kernel_trace.csv")
python tools/step_phases.py $f gpurun_out/r2n/step_phases_384x576.txt
done
