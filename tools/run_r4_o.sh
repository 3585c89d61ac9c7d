cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o
mkdir -p $O
python - > $O/inference.txt 2>&1 <<'PY'
import json, torch, bench, os
dev = torch.device("cuda", 0)
for pf in ("0", "1", "0", "1"):
    os.environ["CDETR_FROZEN_PREFETCH"] = pf
    r = bench.inference_leg(dev, [(800, 800), (384, 576), (800, 800, 8)], 2, "bf16x3")
    print("prefetch", pf, [(s["image"], s["images_per_gpu"], round(s["graph"]["value"], 1), round(s["eager"]["value"], 1)) for s in r["shapes"]])
PY
cat $O/inference.txt | tail -4
python -m pytest tests/test_graph_cache.py -m gpu -x -q -k inference > $O/tests.log 2>&1; tail -2 $O/tests.log
