# round 5, split reductions in the direct-to-LDS tile kernel: parity, in-step A/B (same lease)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5split
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_dl.py -x -q -m gpu -k "split_reduction" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for r in 1 2 3; do for v in 0 1; do
  echo -n "CDETR_DL_SPLITK=$v "; CDETR_DL_SPLITK=$v timeout 300 python bench.py --mode graph --steps 30 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done; done | tee $O/ab_step2.txt
