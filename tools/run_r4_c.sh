# Round 4, run 3: frozen-stage prefetch -- parity tests, same-lease A/B, kernel trace of the pipelined step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
python -m pytest tests/test_graph_cache.py tests/test_gemm_dl.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -m pytest tests/test_model_gpu.py tests/test_dp_shared_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; tail -3 $O/tests2.log
bash tools/run_ab.sh "CDETR_FROZEN_PREFETCH=0" "CDETR_FROZEN_PREFETCH=1" 3 2>&1 | tee $O/ab_prefetch.txt
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 3 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
cp $f $O/kernel_trace.csv
python tools/step_phases.py $f $O/step_phases_800x800.txt > /dev/null 2>&1
cat $O/step_phases_800x800.txt
