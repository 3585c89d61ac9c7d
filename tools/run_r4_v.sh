cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4v
mkdir -p $O
python tools/stride_pad.py > $O/stride_pad.txt 2>&1; cat $O/stride_pad.txt
