cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
python tools/rcda_probe.py 2>&1 | grep -v amdgpu.ids > $O/rcda_probe4.txt; grep "dS kernel" $O/rcda_probe4.txt | cut -c1-700
