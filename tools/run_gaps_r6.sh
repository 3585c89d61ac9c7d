# main-queue idle gaps of a PIPELINED captured step under different submission orders / side-stream settings (tools/step_gaps.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for cfg in "CDETR_B_FIRST=1" "CDETR_B_FIRST=0" "CDETR_B_FIRST=1 CDETR_PF_POST_US=0"; do
  rm -rf /tmp/prof_g; env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python bench.py --mode graph --steps 6 --warmup 2 $F > /tmp/g.log 2>&1
  f=$(find /tmp/prof_g -name "*kernel_trace.csv")
  echo "== $cfg"
  python tools/step_gaps.py $f 15 8
done
STEPS=40 bash tools/run_ab_env.sh gpurun_out/r6_ab_bfirst.txt 3 "CDETR_B_FIRST=1" "CDETR_B_FIRST=0"
