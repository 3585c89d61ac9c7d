# two ranks on ONE GPU (gloo): exercises the N>1 control flow of bench.py / engine.Trainer (hooks, bucketed exchange, graph A / all-reduce / graph B)
cd $GRAFT_REPO_ROOT
export CDETR_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in eager graph auto; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --mode $mode --no-cpu-baseline --no-alt 2>&1 | tail -3 | cut -c1-700
done
