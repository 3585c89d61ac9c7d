"""bf16x3 igemm_fast: per-fragment split (CDETR_GEMM_SPLIT=0) vs split-at-staging (=1) over tile variants.
usage: CDETR_GEMM_SPLIT=0|1 python tools/split_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import run, NAMES
SH = [(5000, 256, 256, 1, 0, None), (5000, 256, 256, 1, 1, None), (5000, 1024, 256, 1, 0, None), (5000, 256, 1024, 1, 1, None),
      (5000, 512, 2048, 1, 0, None), (5000, 2048, 512, 1, 1, None), (5000, 256, 4096, 1, 0, None),
      (5000, 512, 512, 9, 0, (50, 50, 1, 2, 2)), (5000, 512, 512, 9, 1, (50, 50, 1, 2, 2)),
      (20000, 128, 128, 9, 0, (100, 100, 1, 1, 1)), (20000, 128, 128, 9, 1, (100, 100, 1, 1, 1)),
      (20000, 512, 128, 1, 0, None), (20000, 128, 512, 1, 1, None), (80000, 256, 64, 1, 0, None)]
VS = (4, 9, 12, 10, 11, 3)
print("split =", os.environ.get("CDETR_GEMM_SPLIT", "0"))
for sh in SH:
    row = []
    for v in VS:
        if v in (3, 11, 12) and sh[2] % 64:
            row.append("      -     "); continue
        us, tf = run(sh, v)
        row.append(f"{us:7.1f}us {tf:5.1f}")
    print(f"{str(sh[:5]):32s} " + " | ".join(f"{NAMES[v]}: {r}" for v, r in zip(VS, row)), flush=True)
