cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe2.txt; grep "score phase" $O/rcda_probe2.txt
bash tools/run_r5_ab.sh $O/ab_rg.txt 3 "-" "CDETR_RCDA_RG=2"
