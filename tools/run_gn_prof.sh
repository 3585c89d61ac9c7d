cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py -x -q -k "groupnorm or forward_losses or train_step" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -- python bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-alt > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-160
f=$(find /tmp/prof_g -name "*kernel_stats.csv" | head -1)
python tools/kernel_stats.py $f 80 1 | grep -i "gn_fwd\|gn_bwd" | cut -c1-120
