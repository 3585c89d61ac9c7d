cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2t
python tools/wgrad_twin_bench.py 2>&1 | tee gpurun_out/r2t/twin_bench.txt | tail -8
