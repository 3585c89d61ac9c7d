cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
for sk in 1 0 1 0 1 0; do echo "CDETR_RCDA_SKIP_SAVE=$sk  $(CDETR_RCDA_SKIP_SAVE=$sk python tools/infer_ab.py 2>/dev/null | tail -1)"; done > $O/ab_skip_save.txt 2>&1
cat $O/ab_skip_save.txt
