cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
python tools/rcda_probe.py > $O/rcda_probe.txt 2>&1; cat $O/rcda_probe.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "lvis_wide or cfg2" > $O/t_full.log 2>&1; echo "full rc=$?"
timeout 900 python -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -s > $O/t_timed.log 2>&1; echo "timed rc=$?"
grep -h "element-wise\|buckets" $O/t_full.log $O/t_timed.log | cut -c1-400
tail -3 $O/t_full.log; tail -3 $O/t_timed.log
