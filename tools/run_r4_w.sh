# Round 4, run 24: epilogue operands of the direct-to-LDS tile GEMM -- early (registers across the k-loop), late, late behind an early touch of their lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w
mkdir -p $O
python -m pytest tests/test_gemm_dl.py -m gpu -x -q > $O/tests_default.log 2>&1; tail -1 $O/tests_default.log
CDETR_DL_TOUCH=1 python -m pytest tests/test_gemm_dl.py -m gpu -x -q > $O/tests_touch.log 2>&1; tail -1 $O/tests_touch.log
for m in "CDETR_X=0" "CDETR_DL_LATE_EPILOGUE=1" "CDETR_DL_TOUCH=1"; do echo "== $m"; env $m python tools/dl_sweep.py bwd 2>&1 | grep -v amdgpu.ids | cut -c1-230; done > $O/dl_sweep_touch.txt 2>&1
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_X=0" "CDETR_DL_LATE_EPILOGUE=1" "CDETR_DL_TOUCH=1"; do ab "$s"; done
done 2>&1 | tee $O/ab_touch.txt
