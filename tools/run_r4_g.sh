# Round 4, run 7: host-side cost of the graph launches under runtime knobs / graph topologies
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
for s in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "AMD_DIRECT_DISPATCH=0" "CDETR_BRANCH_BESIDE=0 CDETR_WGRAD_EVERY=0" "CDETR_BRANCH_BESIDE=0 CDETR_WGRAD_EVERY=0 CDETR_FROZEN_PREFETCH=0" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_DYNAMIC_QUEUES=0"; do
  echo "== $s"
  env $s python tools/host_cost.py 2>&1 | tail -3
done 2>&1 | tee $O/host_cost.txt
