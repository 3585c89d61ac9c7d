"""Where does a workgroup of the direct-to-LDS tile kernel spend its time?  s_memtime stamps of wave 0 at the phase boundaries
(csrc/igemm_dl.hip PROBE build, written into the split-reduction scratch): prologue issue, first tile landed, k-loop, epilogue
operand loads, stores issued, stores acknowledged.  Cold operands.  usage: python tools/dl_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from counting_detr_amd import ops, _ffi

DEV = "cuda"
SHAPES = [(5000, 2048, 512, 1, None, True), (5000, 1024, 256, 1, None, True), (5000, 256, 256, 9, (50, 50, 1, 1, 1), False),
          (20000, 512, 128, 1, None, True), (80000, 256, 64, 1, None, True), (5000, 512, 2048, 1, None, False)]
names = {0: "128x128", 1: "128x64", 3: "64x64"}
for precision in (1, 3):
    for (M, N, K, taps, geo, resid) in SHAPES:
        g = None
        if geo is not None:
            H, W, stride, pad, dil = geo
            g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
        w4 = torch.randn(N, taps * K, device=DEV) / (K * taps) ** 0.5
        mir = ops.WeightMirror([], [(w4, None)])
        mir.refresh("fwd")
        sp = mir.lookup_fwd(w4)
        w16 = w4.to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        ns = 6
        As = [torch.randn(M, K, device=DEV) for _ in range(ns)]
        planes = [ops.split_planes(a) for a in As]
        Rs = [torch.randn(M, N, device=DEV) for _ in range(ns)]
        Cs = [torch.empty(M, N, device=DEV) for _ in range(ns)]
        Ch = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(ns)]
        Cl = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(ns)]
        ws = torch.zeros(1 << 22, dtype=torch.int64, device=DEV)
        flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
        for tile in (0, 1, 3):
            bm, bn = {0: (128, 128), 1: (128, 64), 3: (64, 64)}[tile]
            nblk = 8 * (((M + bm - 1) // bm + 7) // 8) * ((N + bn - 1) // bn)
            evs = []
            for i in range(ns):
                flush.zero_()
                ws.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm_raw(As[i], K, w4, taps * K, Cs[i], N, M, N, K, taps=taps, bias=bias, relu=True, resid=Rs[i] if resid else None, ldr=N,
                             geom=g, B_split=sp, B16=w16 if precision == 3 else None, precision=precision, A16=planes[i][0], A16lo=planes[i][1] if precision == 1 else None,
                             C16=Ch[i], C16lo=Cl[i] if precision == 1 else None, dl=(tile, 103), probe_ws=ws)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ev_us = evs[-1][0].elapsed_time(evs[-1][1]) * 1e3
            t = ws[:nblk * 8].view(nblk, 8).cpu().numpy().astype(np.float64)
            t = t[(t[:, 0] > 0) & (t[:, 6] > 0)]
            t0 = t[:, 0].min()
            TPU = float(os.environ.get("TICKS_PER_US", "2400"))      # s_memtime ticks per microsecond (printed raw span lets one calibrate against the event time)
            ph = np.diff(t[:, :7], axis=1) / TPU
            med = np.median(ph, axis=0)
            p90 = np.percentile(ph, 90, axis=0)
            span = (t[:, 6].max() - t0) / TPU
            start = (t[:, 0] - t0) / TPU
            print(f"prec {precision} {(M, N, K, taps)} {names[tile]}: {len(t)} WGs, event {ev_us:6.1f} us, span {span:6.1f} us ({t[:, 6].max() - t0:.0f} ticks) | median/p90 us: issue {med[0]:.2f}/{p90[0]:.2f}  "
                  f"first tile {med[1]:.2f}/{p90[1]:.2f}  k-loop {med[2]:.2f}/{p90[2]:.2f}  epi loads {med[3]:.2f}/{p90[3]:.2f}  "
                  f"epi math+store issue {med[4]:.2f}/{p90[4]:.2f}  store ack {med[5]:.2f}/{p90[5]:.2f} | WG start median {np.median(start):.1f} max {start.max():.1f}", flush=True)
