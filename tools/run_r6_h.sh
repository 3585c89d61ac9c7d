cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "rcda" > $O/t_rcda.log 2>&1; echo "rcda rc=$?"; tail -2 $O/t_rcda.log
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe3.txt; grep "score phase\|fwd " $O/rcda_probe3.txt | head -12
python tools/rcda_wide.py 2>&1 | grep -v amdgpu.ids | head -11
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "forward_intermediates" > $O/t_full.log 2>&1; echo "full rc=$?"; tail -2 $O/t_full.log
