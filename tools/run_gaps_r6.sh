# main-queue idle gaps of a PIPELINED captured step (tools/step_gaps.py): with / without the event between F and B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for cfg in "CDETR_PF_TIMEOUT_US=4000" "CDETR_NO_EVF=1 CDETR_PF_TIMEOUT_US=4000"; do
  rm -rf /tmp/prof_g; env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python bench.py --mode graph --steps 6 --warmup 2 $F > /tmp/g.log 2>&1
  f=$(find /tmp/prof_g -name "*kernel_trace.csv")
  echo "== $cfg"
  python tools/step_gaps.py $f 15 20 | grep -v "igemm_dl_kernel<1, 1, 1, 3, false, true, 0, false> and igemm_dl"
  python tools/step_gaps.py $f 14 20 | grep -v "igemm_dl_kernel<1, 1, 1, 3, false, true, 0, false> and igemm_dl"
done
STEPS=40 bash tools/run_ab_env.sh gpurun_out/r6_ab_noevf.txt 3 "CDETR_PF_TIMEOUT_US=4000" "CDETR_NO_EVF=1 CDETR_PF_TIMEOUT_US=4000"
