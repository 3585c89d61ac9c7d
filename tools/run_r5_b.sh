cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
timeout 1200 python -m pytest tests/test_timed_path_gpu.py -x -q -s > $O/tests_timed.log 2>&1; echo "timed-path tests rc=$?"
tail -3 $O/tests_timed.log
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/stagger_probe.py > $O/stagger_q8.txt 2>&1; cat $O/stagger_q8.txt | grep -v "^W2026\|amdgpu.ids"
timeout 600 python tools/stagger_probe.py > $O/stagger_q4.txt 2>&1; cat $O/stagger_q4.txt | grep -v "^W2026\|amdgpu.ids"
