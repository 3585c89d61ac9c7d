cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
for s in "CDETR_SPLITK=0" "CDETR_SPLITK=1 CDETR_GEMM_FEWROW_SPLIT=0" "CDETR_SPLITK=1" "CDETR_SPLITK=0" "CDETR_SPLITK=1"; do
  env $s python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$s', round(d['ms_per_step'],3), d['step_ms']['median'], d['config']['final_loss'])"
done
