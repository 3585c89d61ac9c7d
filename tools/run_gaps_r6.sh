# main-queue idle gaps of a PIPELINED captured step (tools/step_gaps.py) under prefetch-release variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for cfg in "CDETR_NOTHING=1" "CDETR_PF_TIMEOUT_US=0" "CDETR_PF_POST_US=0" "CDETR_PF_POST_US=100" "GPU_MAX_HW_QUEUES=2"; do
  rm -rf /tmp/prof_g; env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python bench.py --mode graph --steps 6 --warmup 2 $F > /tmp/g.log 2>&1
  f=$(find /tmp/prof_g -name "*kernel_trace.csv")
  echo "== $cfg"
  python tools/step_gaps.py $f 15 20 | grep -v "igemm_dl_kernel<1, 1, 1, 3, false, true, 0, false> and igemm_dl"
done
