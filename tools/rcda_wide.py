"""RCDA core at the FSCD-LVIS map sizes (BASELINE configs[3]: images up to 800 x 1333 -> 50 x 84 keys at stride 16, or 84 x 50 for a
portrait image) against the square 50 x 50 map of configs[1]: HIP-event time per call and algorithmic TF, with the round-6 wide-map
kernels (rcda_fwd2_kernel<4, 5|6> / <.., TH = 3>, rcda_dv2 key-column chunks) and with CDETR_RCDA_WIDE=0 (the rounds 1-5 dispatch: W > 64
falls back to rcda_fwd_kernel / rcda_dv_kernel).  -> profiles/r6_rcda_wide.txt"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from counting_detr_amd import ops
    nh, E, N, dev = 8, 256, 2, "cuda"

    def timeit(fn, reps=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for H, W in ((50, 50), (50, 84), (84, 50), (50, 64), (50, 96)):
        for L in (H * W, 300):
            g = torch.Generator(device=dev).manual_seed(1)
            mk = lambda *s: torch.randn(*s, device=dev, generator=g)
            q_row, q_col, k_row, k_col, v, dO = mk(N, L, E), mk(N, L, E), mk(N, W, E), mk(N, H, E), mk(N, H, W, E), mk(N, L, E)
            o, a_row, a_col = ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh)
            zb = torch.zeros(ops.rcda_zero_numel(v, k_row, k_col), device=dev)
            tf = timeit(lambda: ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh))
            tb = timeit(lambda: ops.rcda_bwd_raw(dO, q_row, q_col, k_row, k_col, v, a_row, a_col, nh, zbuf=zb))
            fl = 2.0 * N * nh * L * (H * W * 32 + (H + W) * 32)
            print("H=%3d W=%3d L=%5d  fwd %7.1f us %6.1f TF   bwd %7.1f us %6.1f TF (3 products)" % (H, W, L, tf, fl / tf * 1e-6, tb, 3 * fl / tb * 1e-6), flush=True)


if __name__ == "__main__":
    if os.environ.get("RCDA_WIDE_CHILD"):
        child()
    else:
        for wide in ("1", "0"):
            print("== CDETR_RCDA_WIDE=%s (%s)" % (wide, "round 6 dispatch" if wide == "1" else "rounds 1-5 dispatch: W > 64 on the fallback kernels"), flush=True)
            env = dict(os.environ, RCDA_WIDE_CHILD="1", CDETR_RCDA_WIDE=wide)
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
