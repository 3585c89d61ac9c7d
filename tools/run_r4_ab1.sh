# Round 4, run 2: parity of the changed kernels, then same-lease A/Bs: epilogue operands requested before the k-loop (igemm_dl) and
# twin-only inner gradients; phase probe of the direct-to-LDS kernel.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b
mkdir -p $O
python -m pytest tests/test_gemm_dl.py tests/test_full_size_gpu.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -x -q -k "conv or bottleneck or trunk or backbone or trajectory or adamw" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
bash tools/run_ab.sh "CDETR_DL_LATE_EPILOGUE=1" "CDETR_X=0" 3 2>&1 | tee $O/ab_hoist.txt
bash tools/run_ab.sh "CDETR_TWIN_ONLY=0" "CDETR_TWIN_ONLY=1" 3 2>&1 | tee $O/ab_twin_only.txt
python tools/dl_probe.py > $O/dl_probe_early.txt 2>&1
CDETR_DL_LATE_EPILOGUE=1 python tools/dl_probe.py > $O/dl_probe_late.txt 2>&1
grep "64x64" $O/dl_probe_early.txt | cut -c1-330 | head -12
