"""Direct-to-LDS tile kernel (csrc/igemm_dl.hip) vs the register-staged tile kernels on the backbone's GEMM shapes at two images
800x800 per GPU, COLD operands (a ring of buffer sets larger than L2 + Infinity Cache), every tile / ring-depth configuration.
The register-staged kernel gets what it gets in the step (fp32 A + pre-split B, plus the bf16 twin of A for plain bf16); the
direct-to-LDS kernel the pre-split planes.  Both write C (fp32) + C16 (+ C16lo for the forward).
usage: python tools/dl_sweep.py [fwd|bwd|all] [M-multiplier]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi

DEV = "cuda"
GROUPS = os.environ.get("DL_SWEEP_GROUPS", "0") == "1"      # also time the interleaved-groups operand format (forward only)
CONFIGS = [(0, 2), (0, 3), (1, 2), (1, 3), (1, 4), (2, 3), (3, 3), (3, 4)]
if os.environ.get("DL_SWEEP_SHALLOW") == "1":      # round 5: 2-deep rings on the small tiles (32 KB of LDS: 5 workgroups per CU instead of 3)
    CONFIGS = [(1, 2), (1, 3), (2, 2), (2, 3), (3, 2), (3, 3)]
if os.environ.get("DL_SWEEP_SPLIT") == "1":        # round 5: split reductions (stages code 200 + 10 * slices + ring depth) next to the unsplit tiles
    CONFIGS = [(0, 3), (2, 3), (3, 3), (3, 223), (3, 233), (3, 243), (3, 222), (3, 242), (2, 223), (2, 233), (0, 223), (0, 233)]
if os.environ.get("DL_SWEEP_ONLY"):                # comma-separated indices into FWD
    _only = [int(x) for x in os.environ["DL_SWEEP_ONLY"].split(",")]
# (M rows, N, K, taps, conv geometry (H, W, stride, pad, dil) or None, epilogue with residual)
FWD = [(80000, 64, 64, 1, None, False), (80000, 64, 64, 9, (200, 200, 1, 1, 1), False), (80000, 256, 64, 1, None, True),
       (80000, 64, 256, 1, None, False), (20000, 128, 256, 1, None, False), (20000, 128, 128, 9, (100, 100, 1, 1, 1), False),
       (20000, 512, 128, 1, None, True), (20000, 128, 512, 1, None, False), (5000, 256, 512, 1, None, False),
       (5000, 256, 256, 9, (50, 50, 1, 1, 1), False), (5000, 1024, 256, 1, None, True), (5000, 256, 1024, 1, None, False),
       (5000, 512, 1024, 1, None, False), (5000, 512, 512, 9, (50, 50, 1, 2, 2), False), (5000, 2048, 512, 1, None, True),
       (5000, 512, 2048, 1, None, False)]


def bench(call, nsets, reps=None):
    for i in range(nsets):
        call(i)
    torch.cuda.synchronize()
    reps = reps or max(2 * nsets, 12)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        call(r % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(shape, precision, mult=1):
    M, N, K, taps, geo, resid = shape
    M *= mult
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    w4 = (torch.randn(N, taps * K, device=DEV) / (K * taps) ** 0.5)
    mir = ops.WeightMirror([], [(w4, None)])
    mir.refresh("fwd")
    sp = mir.lookup_fwd(w4)
    w16 = w4.to(torch.bfloat16)          # the plain-bf16 weight image (cdetr_gemm_desc.B16)
    bias = torch.randn(N, device=DEV)
    per_set = 4 * (M * K * 2 + M * N * 3)
    nsets = max(2, min(16, int(1.5e9 // per_set)))
    As = [torch.randn(M, K, device=DEV) for _ in range(nsets)]
    Ah, Al = zip(*[ops.split_planes(a) for a in As])
    Rs = [torch.randn(M, N, device=DEV) for _ in range(nsets)] if resid else None
    Cs = [torch.empty(M, N, device=DEV) for _ in range(nsets)]
    C16 = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(nsets)]
    C16l = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(nsets)] if precision == 1 else None

    def old(i):
        ops.gemm_raw(As[i], K, w4, taps * K, Cs[i], N, M, N, K, taps=taps, bias=bias, relu=True, resid=Rs[i] if resid else None, ldr=N, geom=g,
                     B_split=sp, precision=precision, A16=Ah[i] if precision == 3 else None, C16=C16[i],
                     C16lo=C16l[i] if (C16l and not GROUPS) else None)      # (GROUPS: what the product's forward writes: fp32 + the hi twin = 6 bytes per element, like the grouped form)
    row = {}
    os.environ["CDETR_GEMM_DL"] = "0"
    row["old"] = bench(old, nsets)
    if GROUPS and precision == 1:
        # round 4: the activation stored ONCE as interleaved groups [hi 32 | lo 32] (the format of B_split) instead of fp32 + planes: A read
        # as full 128-byte lines per k-tile, C written as groups (+ the plain hi twin the weight gradients read), no fp32 tensor at all
        Ag = [ops.split_groups(a) for a in As]
        Cg = [torch.empty(M, N // 32, 64, device=DEV, dtype=torch.bfloat16) for _ in range(nsets)] if N % 32 == 0 else None
        for tile, stages in CONFIGS:
            def grp(i):
                ops.gemm_raw(None, K, w4, taps * K, None if Cg else Cs[i], N, M, N, K, taps=taps, bias=bias, relu=True, resid=Rs[i] if resid else None, ldr=N,
                             geom=g, B_split=sp, precision=1, A_split=Ag[i], C16=C16[i], C_split=Cg[i] if Cg else None, dl=(tile, stages))
            row[("g", tile, stages)] = bench(grp, nsets)
    for tile, stages in CONFIGS:
        def new(i):
            ops.gemm_raw(As[i], K, w4, taps * K, Cs[i], N, M, N, K, taps=taps, bias=bias, relu=True, resid=Rs[i] if resid else None, ldr=N, geom=g,
                         B_split=sp, B16=w16 if precision == 3 else None, precision=precision, A16=Ah[i], A16lo=Al[i] if precision == 1 else None, C16=C16[i],
                         C16lo=C16l[i] if C16l else None, dl=(tile, stages))
        if precision == 3 and K % 64:
            continue
        if stages >= 200 and ((stages - 200) // 10 > K * taps // (64 if precision == 3 else 32) or M * N > (1 << 22)):
            continue
        row[(tile, stages)] = bench(new, nsets)
    return row, 2.0 * M * N * K * taps


if __name__ == "__main__":
    os.environ["CDETR_GEMM_DL"] = "0"          # cdetr_gemm itself stays on the register-staged kernels (read once at first call)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    mult = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    names = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}
    for precision, tag in ((1, "fwd"), (3, "bwd")):
        if what not in ("all", tag):
            continue
        print(f"== {tag}: precision {precision} ({'bf16x3' if precision == 1 else 'bf16'}), M x{mult}; us per launch (TF algorithmic)")
        print("%-34s %14s | " % ("M N K taps", "reg-staged") + " ".join("%13s" % f"{names[t]}/{s}" for t, s in CONFIGS))
        tot_old = tot_best = tot_g = tot_gonly = 0.0
        for si, sh in enumerate(FWD):
            if os.environ.get("DL_SWEEP_ONLY") and si not in _only:
                continue
            row, fl = run(sh, precision, mult)
            best = min((v, k) for k, v in row.items() if k != "old" and k[0] != "g")
            tot_old += row["old"]
            tot_best += min(best[0], row["old"])
            cells = " ".join(("%7.1f (%4.0f)" % (row[c], fl / row[c] / 1e6)) if c in row else "%13s" % "-" for c in CONFIGS)
            print("%-34s %7.1f (%4.0f) | %s   best %s/%d x%.2f" % (str(sh[:4]), row["old"], fl / row["old"] / 1e6, cells, names[best[1][0]], best[1][1],
                                                                  row["old"] / best[0]), flush=True)
            if any(k[0] == "g" for k in row if k != "old"):
                cells = " ".join(("%7.1f (%4.0f)" % (row[("g",) + c], fl / row[("g",) + c] / 1e6)) if ("g",) + c in row else "%13s" % "-" for c in CONFIGS)
                bg = min((v, k) for k, v in row.items() if k != "old" and k[0] == "g")
                tot_g += min(bg[0], row["old"])
                tot_gonly += bg[0]
                print("%-34s %14s | %s   best %s/%d x%.2f vs reg-staged, x%.2f vs planes" % ("   ... interleaved groups", "", cells, names[bg[1][1]], bg[1][2],
                                                                                             row["old"] / bg[0], best[0] / bg[0]), flush=True)
        print(f"sum: reg-staged {tot_old:.0f} us, best-of {tot_best:.0f} us" + (f", interleaved groups: best-of-with-reg-staged {tot_g:.0f} us, groups everywhere {tot_gonly:.0f} us" if tot_g else ""))
