cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-alt"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc1 -- $B > /dev/null 2>&1
f=$(find /tmp/pc1 -name "*counter_collection.csv"); python tools/pmc_summary.py $f "$1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pc2 -- $B > /dev/null 2>&1
f=$(find /tmp/pc2 -name "*counter_collection.csv"); python tools/pmc_summary.py $f "$1"
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/pc3 -- $B > /dev/null 2>&1
f=$(find /tmp/pc3 -name "*counter_collection.csv"); python tools/pmc_summary.py $f "$1"
