"""Where does a wave of the two-step RCDA forward (rcda_fwd2_kernel, csrc/rcda.hip PROBE build) spend its time?  s_memtime readings of
every wave: score phase, main loop (50 barrier-separated key rows), store; inside the loop the cycles spent waiting at the per-key-row
barrier, in the MFMA + accumulate section and in global fetch + LDS stash.  Encoder shape (2 x 8 heads x 2500 queries, 50 x 50 keys)
with 5-wave and 4-wave workgroups, decoder shape (300 queries).  usage: python tools/rcda_probe.py  -> profiles/r6_rcda_probe.txt"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(nw):
    import numpy as np
    import torch
    from counting_detr_amd import ops
    N, nh, E, H, W, dev = 2, 8, 256, 50, 50, "cuda"
    TPU = float(os.environ.get("TICKS_PER_US", "2100"))           # s_memtime ticks per microsecond: it counts shader-clock cycles here (first run: 111 000 ticks per 53 us wave)
    for L in (H * W, 300):
        g = torch.Generator(device=dev).manual_seed(1)
        mk = lambda *s: torch.randn(*s, device=dev, generator=g)
        q_row, q_col, k_row, k_col, v = mk(N, L, E), mk(N, L, E), mk(N, W, E), mk(N, H, E), mk(N, H, W, E)
        ws = ops.splitk_ws()
        evs = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ev_us = evs[-1][0].elapsed_time(evs[-1][1]) * 1e3
        nwg = ((L + 32 * nw - 1) // (32 * nw)) * N * nh
        t = ws.view(torch.int64)[:nwg * nw * 16].view(nwg * nw, 16).cpu().numpy().astype(np.float64)
        ws.zero_()                                               # (the stamps overwrote the arrival counters at the head of the scratch)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        span = (t[:, 4].max() - t0) / TPU
        ph = np.diff(t[:, :5], axis=1) / TPU
        inner = t[:, 5:8] / TPU
        med, p90 = np.median(ph, axis=0), np.percentile(ph, 90, axis=0)
        imed = np.median(inner, axis=0)
        start = (t[:, 0] - t0) / TPU
        nh_ = H
        sc = np.median(np.diff(np.concatenate([t[:, 0:1], t[:, 8:14]], axis=1), axis=1) / TPU, axis=0)
        print(f"NW={nw} L={L}: score phase, median us per wave: key + query loads landed {sc[0]:.2f} | barrier {sc[1]:.2f} | row side (MFMA logits, softmax, LDS) {sc[2]:.2f} | "
              f"column side {sc[3]:.2f} | barrier {sc[4]:.2f} | saving both maps {sc[5]:.2f}", flush=True)
        print(f"NW={nw} L={L}: {nwg} workgroups x {nw} waves, event {ev_us:.1f} us (probe build), span {span:.1f} us ({t[:, 4].max() - t0:.0f} ticks) | per wave, median / p90 us: "
              f"scores {med[0]:.2f}/{p90[0]:.2f}  A_row split + first tiles {med[1]:.2f}/{p90[1]:.2f}  main loop {med[2]:.2f}/{p90[2]:.2f}  store {med[3]:.2f}/{p90[3]:.2f} | "
              f"inside the loop (sum over {nh_} key rows, median): barrier wait {imed[0]:.2f}  MFMA + accumulate {imed[1]:.2f}  fetch + stash {imed[2]:.2f}  "
              f"= per key row {imed[0] / nh_ * 1e3:.0f} + {imed[1] / nh_ * 1e3:.0f} + {imed[2] / nh_ * 1e3:.0f} ns | wave start median {np.median(start):.1f} max {start.max():.1f} us", flush=True)


def child_bwd(nw):
    """The dS / dq / dk kernel (rcda_bwd_kernel<2, nw, 1, 1>, plain-bf16 products): the PROBE build takes its stamp buffer through ds_row."""
    import ctypes as C
    import numpy as np
    import torch
    from counting_detr_amd import ops
    from counting_detr_amd._ffi import RcdaBwdDesc, check, lib, ptr, stream_ptr
    N, nh, E, H, W, dev = 2, 8, 256, 50, 50, "cuda"
    TPU = float(os.environ.get("TICKS_PER_US", "2100"))
    for L in (H * W, 300):
        g = torch.Generator(device=dev).manual_seed(1)
        mk = lambda *s: torch.randn(*s, device=dev, generator=g)
        q_row, q_col, k_row, k_col, v, dO = mk(N, L, E), mk(N, L, E), mk(N, W, E), mk(N, H, E), mk(N, H, W, E), mk(N, L, E)
        os.environ.pop("CDETR_RCDA_PROBE", None)      # (the forward below is the product kernel; its static was read at the first call anyway)
        o, a_row, a_col = ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh, save=True)
        Hp, Wp = ops.rcda_pads(H, W)
        nwg = ((L + 32 * nw - 1) // (32 * nw)) * N * nh
        stamps = torch.zeros(nwg * nw * 16, dtype=torch.int64, device=dev)
        dummy = torch.empty_like(a_col)
        zb = torch.zeros(ops.rcda_zero_numel(v, k_row, k_col), device=dev)
        d_v = zb[:v.numel()].view(v.shape)
        dk = zb[v.numel():]
        dk_row, dk_col = dk[:k_row.numel()].view(k_row.shape), dk[k_row.numel():].view(k_col.shape)
        dq_row, dq_col = torch.empty_like(q_row), torch.empty_like(q_col)
        d = RcdaBwdDesc()
        d.N, d.L, d.H, d.W, d.nh, d.scale, d.precision = N, L, H, W, nh, 32 ** -0.5, 3
        d.d_out, d.a_row, d.a_col, d.v, d.d_v = ptr(dO), ptr(a_row), ptr(a_col), ptr(v), ptr(d_v)
        d.k_row, d.k_col, d.dq_row, d.dq_col = ptr(k_row), ptr(k_col), ptr(dq_row), ptr(dq_col)
        d.q_row, d.q_col, d.dk_row, d.dk_col = ptr(q_row), ptr(q_col), ptr(dk_row), ptr(dk_col)
        d.ds_row, d.ds_col = ptr(stamps), ptr(dummy)
        os.environ["CDETR_RCDA_PROBE"] = str(nw)      # read once, at the first cdetr_rcda_bwd call
        evs = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().cdetr_rcda_bwd(C.byref(d), stream_ptr()), "cdetr_rcda_bwd")
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = stamps.view(nwg * nw, 16).cpu().numpy().astype(np.float64)
        t = t[t[:, 0] > 0]
        ph = np.diff(t[:, :7], axis=1) / TPU
        med, p90 = np.median(ph, axis=0), np.percentile(ph, 90, axis=0)
        names = ["attention rows staged, A_col in registers", "main loop (%d key columns)" % W, "both softmax backward passes", "keys -> LDS + barrier",
                 "dq (MFMA) + store", "dk (MFMA) + atomics"]
        print(f"dS kernel NW={nw} L={L}: {nwg} workgroups, event {evs[-1][0].elapsed_time(evs[-1][1]) * 1e3:.1f} us (probe build, dS launch only) | median / p90 us per wave: " +
              " | ".join(f"{n_} {m:.2f}/{p:.2f}" for n_, m, p in zip(names, med, p90)) + f" | total {np.median(t[:, 6] - t[:, 0]) / TPU:.1f}", flush=True)


if __name__ == "__main__":
    if os.environ.get("RCDA_PROBE_BWD"):
        child_bwd(int(os.environ["CDETR_RCDA_PROBE"]))
    elif os.environ.get("CDETR_RCDA_PROBE"):
        child(int(os.environ["CDETR_RCDA_PROBE"]))
    else:
        for rg in (1, 2):
            print(f"== key rows per barrier (CDETR_RCDA_RG) = {rg}", flush=True)
            for nw in (5, 4):
                subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CDETR_RCDA_PROBE=str(nw), CDETR_RCDA_RG=str(rg)))
        print("== backward: dS / dq / dk kernel", flush=True)
        for nw in (5, 4):
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CDETR_RCDA_PROBE=str(nw), RCDA_PROBE_BWD="1"))
        print("== un-instrumented kernels, HIP events (tools/rcda_time.py)", flush=True)
        for rg in (1, 2):
            print(f"CDETR_RCDA_RG={rg}", flush=True)
            subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rcda_time.py")], env=dict(os.environ, CDETR_RCDA_RG=str(rg)))
