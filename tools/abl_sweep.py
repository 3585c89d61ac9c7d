"""igemm_fast 64x64 BK32 bf16x3 ablations: CDETR_GEMM_ABL = 0 full, 3 = B operand staged without split, 4 = neither split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import run
SH = [(5000, 256, 256, 1, 0, None), (5000, 1024, 256, 1, 0, None), (5000, 512, 2048, 1, 0, None), (5000, 256, 1024, 1, 0, None),
      (5000, 512, 512, 9, 0, (50, 50, 1, 2, 2)), (20000, 128, 128, 9, 0, (100, 100, 1, 1, 1)), (20000, 512, 128, 1, 0, None)]
print("ABL =", os.environ.get("CDETR_GEMM_ABL", "0"))
for sh in SH:
    us, tf = run(sh, 4)
    print(f"{str(sh[:5]):32s} {us:7.1f}us {tf:6.1f} TF", flush=True)
