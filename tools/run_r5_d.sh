cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
timeout 600 python tools/halo_sweep.py > $O/halo_sweep.txt 2>&1; grep -v "^W2026\|amdgpu.ids" $O/halo_sweep.txt
timeout 600 python -m pytest tests/test_timed_path_gpu.py -x -q -k exchange > $O/tests_exch.log 2>&1; echo "exchange test rc=$?"; tail -3 $O/tests_exch.log
