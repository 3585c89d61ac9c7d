"""Run one igemm shape repeatedly (for rocprofv3 --pmc).  usage: one_gemm.py M N K taps layout variant [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_sweep import run
M, N, K, taps, bl, variant = [int(x) for x in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
geo = (50, 50, 1, 2, 2) if taps == 9 else None
us, tf = run((M, N, K, taps, bl, geo), variant, reps)
print(f"{M}x{N}x{K}x{taps} layout {bl} variant {variant}: {us:.1f} us {tf:.1f} TF")
