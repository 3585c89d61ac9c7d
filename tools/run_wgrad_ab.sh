cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 1; do
echo "=== TRB=$v"
CDETR_WGRAD_TRB=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pa$v -- python tools/wgrad_pmc.py > /dev/null 2>&1
f=$(find /tmp/pa$v -name "*counter_collection.csv"); python tools/pmc_summary.py $f "64, 64"
CDETR_WGRAD_TRB=$v rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pb$v -- python tools/wgrad_pmc.py > /dev/null 2>&1
f=$(find /tmp/pb$v -name "*counter_collection.csv"); python tools/pmc_summary.py $f "64, 64"
done
