"""Key-row slices of the two-step RCDA forward (cdetr_rcda_fwd_desc.ws): HIP-event time per launch at the encoder (L = H*W) and decoder
(L = 300) shapes for every slice count and both workgroup widths.  CDETR_TUNING=1 is needed (the knobs are re-read per call)."""
import os, sys
os.environ["CDETR_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
N, nh, E = 2, 8, 256
dev = "cuda"
ops.PRECISION = 1


def timeit(fn, reps=20, replays=10):
    """20 launches captured in one HIP graph (the Python call costs more than the kernel), replayed 10 times"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e3


for (H, W) in ((50, 50), (24, 36)):
    for L in (H * W, 300, 576):
        q_row, q_col = torch.randn(N, L, E, device=dev), torch.randn(N, L, E, device=dev)
        k_row, k_col = torch.randn(N, W, E, device=dev), torch.randn(N, H, E, device=dev)
        v = torch.randn(N, H, W, E, device=dev)
        row = []
        for nw in (4, 2):
            os.environ["CDETR_RCDA_NW"] = str(nw)
            for nw5 in ((1, 0) if nw == 4 else (0,)):
                os.environ["CDETR_RCDA_NW5"] = str(nw5)
                for hs in (1, 2, 3, 4, 6):
                    os.environ["CDETR_RCDA_HS"] = str(hs)
                    t = timeit(lambda: ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh))
                    row.append("nw%d%s/hs%d %5.1f" % (nw, "+5" if nw5 else "", hs, t))
        os.environ.pop("CDETR_RCDA_HS")
        os.environ.pop("CDETR_RCDA_NW")
        os.environ.pop("CDETR_RCDA_NW5", None)
        t = timeit(lambda: ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh))
        print("H=%d W=%d L=%d  default %5.1f us | %s" % (H, W, L, t, "  ".join(row)), flush=True)
