# Round 4, run 21: register-staged operand ring of the tile GEMM vs the direct-to-LDS ring
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4u
mkdir -p $O
python -m pytest tests/test_gemm_dl.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DL_SWEEP_REG=1 python tools/dl_sweep.py all > $O/dl_sweep_reg.txt 2>&1; cat $O/dl_sweep_reg.txt | cut -c1-400
