# Round-6 verification on one GPU box: the whole GPU suite, the two-rank rehearsal (gloo, one GPU) of bench.py --gpus 2 incl. the exchange-variant A/B.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6checks; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
bash tools/run_rehearsal.sh > $O/rehearsal.txt 2>&1; echo "rehearsal rc=$?"; grep -c '"metric"' $O/rehearsal.txt; grep "exchange variant" $O/rehearsal.txt | head -8
