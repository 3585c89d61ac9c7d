"""Inference forward (engine.InferenceEngine, graph replay) at the bench's shapes; prints images/s per shape.  A/B knobs through the environment
(e.g. CDETR_RCDA_SKIP_SAVE=0).  usage: python tools/infer_ab.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
r = bench.inference_leg(torch.device("cuda", 0), [(800, 800), (384, 576), (800, 800, 8), (800, 1333)], 2, "bf16x3", steps=steps)
print(" | ".join("%dx%d B=%d: %.1f img/s (%.3f ms)" % (s["image"][0], s["image"][1], s["images_per_gpu"], s["graph"]["value"], s["graph"]["ms_per_batch"]) for s in r["shapes"]), flush=True)
