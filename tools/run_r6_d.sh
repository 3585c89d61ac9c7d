cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
for nwx in 0 6 7 8; do for nwxb in 0 6 8; do
  echo "== CDETR_RCDA_NWX=$nwx CDETR_RCDA_NWXB=$nwxb"; CDETR_RCDA_NWX=$nwx CDETR_RCDA_NWXB=$nwxb python tools/rcda_time.py 2>&1 | grep -v amdgpu.ids
done; done > $O/rcda_nw_sweep.txt 2>&1
cat $O/rcda_nw_sweep.txt
CDETR_RCDA_NWX=8 CDETR_RCDA_NWXB=8 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "rcda" > $O/t_rcda_nw8.log 2>&1; echo "rcda nw8 rc=$?"; tail -2 $O/t_rcda_nw8.log
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "train_step" > $O/t_full.log 2>&1; echo "full rc=$?"
timeout 900 python -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -s > $O/t_timed.log 2>&1; echo "timed rc=$?"
grep -h "element-wise samples\|buckets" $O/t_full.log $O/t_timed.log | sed 's/.*element-wise samples/element-wise samples/' | cut -c1-300
tail -3 $O/t_full.log; tail -3 $O/t_timed.log
