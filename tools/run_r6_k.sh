cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "rcda" > $O/t_rcda.log 2>&1; echo "rcda rc=$?"; tail -2 $O/t_rcda.log
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe6.txt; grep "dS kernel\|^L=" $O/rcda_probe6.txt | cut -c1-700
for k in 0 1 0 1; do echo "CDETR_RCDA_DK_PER_WAVE=$k $(CDETR_RCDA_DK_PER_WAVE=$k python tools/rcda_time.py 2>&1 | grep -v amdgpu | tr '\n' ' ')"; done
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "forward_intermediates or train_step or trajectory or fused" > $O/t_full.log 2>&1; echo "full rc=$?"; tail -2 $O/t_full.log
