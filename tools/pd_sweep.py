"""A/B of the igemm register ring depth (CDETR_GEMM_PD, read once per process): times the auto-selected variant on every
GEMM / conv shape of the step.  usage: CDETR_GEMM_PD=4 python tools/pd_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import run, SHAPES
from counting_detr_amd import ops
ops.PRECISION = 1
for sh in SHAPES:
    us, tf = run(sh, 0, reps=30)
    print("%-34s %8.1fus %6.1fTF" % (",".join(str(x) for x in sh[:5]), us, tf), flush=True)
