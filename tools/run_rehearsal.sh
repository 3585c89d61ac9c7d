# two ranks on ONE GPU (gloo): exercises the N>1 control flow of bench.py / engine.Trainer (hooks, bucketed exchange, chain of linear graphs with the
# all-reduces between the pieces); the second form is bench.py launching its own ranks (python bench.py --gpus 2)
cd $GRAFT_REPO_ROOT
export CDETR_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
for mode in eager graph auto; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --mode $mode $F 2>&1 | tail -2 | cut -c1-600
done
echo "== self-launch"
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 $F 2>&1 | tail -2 | cut -c1-900
