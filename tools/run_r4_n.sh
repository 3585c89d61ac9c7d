cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4n
mkdir -p $O
python -m pytest tests/test_graph_cache.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
ab() { env $1 python bench.py --mode graph --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'inline', round(d.get('frozen_stage_prefetch',{}).get('in_line_ms_per_step',0),3))"; }
for i in 1 2 3; do
  for s in "CDETR_Z_LATE=0" "CDETR_Z_LATE=1"; do ab "$s"; done
done 2>&1 | tee $O/ab_zlate.txt
python - > $O/inference.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
for pf in ("0", "1"):
    import os
    os.environ["CDETR_FROZEN_PREFETCH"] = pf
    r = bench.inference_leg(dev, [(800, 800), (384, 576), (800, 800, 8)], 2, "bf16x3")
    print("prefetch", pf, [(s["image"], s["images_per_gpu"], round(s["graph"]["value"], 1), round(s["eager"]["value"], 1)) for s in r["shapes"]])
PY
cat $O/inference.txt | tail -3
