cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe.txt; cat $O/rcda_probe.txt | cut -c1-900
CDETR_RCDA_RG=2 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "rcda" > $O/t_rcda_rg2.log 2>&1; echo "rcda rg2 rc=$?"; tail -2 $O/t_rcda_rg2.log
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "train_step" > $O/t_full.log 2>&1; echo "full rc=$?"
timeout 900 python -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -s > $O/t_timed.log 2>&1; echo "timed rc=$?"
grep -h "element-wise samples\|buckets" $O/t_full.log $O/t_timed.log | sed 's/.*element-wise samples/element-wise samples/' | cut -c1-300
tail -3 $O/t_full.log; tail -3 $O/t_timed.log
