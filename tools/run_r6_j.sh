cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "rcda" > $O/t_rcda.log 2>&1; echo "rcda rc=$?"; tail -2 $O/t_rcda.log
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe5.txt; grep "dS kernel\|^L=" $O/rcda_probe5.txt | cut -c1-700
