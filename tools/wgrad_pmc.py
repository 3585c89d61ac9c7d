"""Run a few wgrad_fast shapes (for rocprofv3 --pmc / --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import runw
for sh in [(5000, 256, 256, 1, None), (5000, 1024, 256, 1, None), (5000, 512, 2048, 1, None), (5000, 512, 512, 9, (50, 50, 1, 2, 2)),
           (20000, 128, 128, 9, (100, 100, 1, 1, 1)), (20000, 512, 128, 1, None)]:
    us, tf = runw(sh, 0, reps=5)
    print(sh[:4], "%.1f us %.1f TF" % (us, tf))
