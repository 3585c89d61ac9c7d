cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "lvis_wide or cfg2" > $O/t_full.log 2>&1; echo "full rc=$?"
timeout 900 python -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -s > $O/t_timed.log 2>&1; echo "timed rc=$?"
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "wgrad or lsap or rcda" > $O/t_kern.log 2>&1; echo "kern rc=$?"
timeout 900 python bench.py --no-cpu-baseline --no-alt --no-inference --no-real-data > $O/bench_a.log 2> $O/bench_a.err; echo "bench rc=$?"
bash tools/run_rehearsal.sh > $O/rehearsal.txt 2>&1; echo "rehearsal rc=$?"
tail -3 $O/t_full.log; tail -3 $O/t_timed.log; tail -3 $O/t_kern.log
