"""Halo-resident 3x3 form of the direct-to-LDS tile kernel (csrc/igemm_dl.hip, NHS > 0) against what the step runs today on the backbone's
stride-1 3x3 shapes at two 800x800 images, COLD operands (ring of buffer sets larger than L2 + Infinity Cache):
  forward (split-bf16 x3): register-staged tile kernel on fp32 rows (the product's choice) | classic direct-to-LDS from planes | halo form, 4 tiles
  data gradient (plain bf16): classic direct-to-LDS from the twin (the product's choice)   | halo form, 4 tiles
usage: python tools/halo_sweep.py [M-multiplier]      Math: A2/models/resnet.py:146-148 (+ dilation A2/models/backbone.py:153-155)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi

DEV = "cuda"
# (images, H, W, channels, dilation, calls per step forward, calls per step backward)
SHAPES = [(2, 200, 200, 64, 1, 3, 0), (2, 100, 100, 128, 1, 3, 3), (2, 50, 50, 256, 1, 5, 5), (2, 50, 50, 512, 1, 1, 1), (2, 50, 50, 512, 2, 2, 2)]
TILES = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}


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


def run(shape, precision, mult):
    Nb, H, W, Cc, dil, _, _ = shape
    Nb *= mult
    M, N, K = Nb * H * W, Cc, Cc
    mode = _ffi.ROWS_CONV_FWD if precision == 1 else _ffi.ROWS_CONV_DGRAD
    g = _ffi.ConvGeom(mode, H, W, H, W, 3, 3, 1, dil, dil)
    w4 = (torch.randn(N, 9 * K, device=DEV) / (9 * K) ** 0.5)
    mir = ops.WeightMirror([], [(w4, None)])
    mir.refresh("fwd")
    sp = mir.lookup_fwd(w4)
    w16 = w4.to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    per_set = 4 * (M * K * 2 + M * N * 3)
    nsets = max(2, min(16, int(1.5e9 // per_set)))
    As = [torch.randn(Nb, H, W, K, device=DEV) for _ in range(nsets)]
    Ah, Al = zip(*[ops.split_planes(a) for a in As])
    Gt = [torch.randn(Nb, H, W, N, device=DEV).bfloat16() for _ in range(nsets)]
    Cs = [torch.empty(Nb, H, W, N, device=DEV) for _ in range(nsets)]
    C16 = [torch.empty(Nb, H, W, N, device=DEV, dtype=torch.bfloat16) for _ in range(nsets)]
    row = {}
    if precision == 1:
        def old(i):        # what conv2 of a bottleneck runs today: fp32 rows, split at staging; writes fp32 + hi twin (+ lo plane under expand_planes: not charged)
            ops.gemm_raw(As[i], K, w4, 9 * K, Cs[i], N, M, N, K, taps=9, bias=bias, relu=True, geom=g, B_split=sp, precision=1, C16=C16[i])
        os.environ["CDETR_GEMM_DL"] = "0"
        row["old"] = bench(old, nsets)
        os.environ["CDETR_GEMM_DL"] = "1"

        def mk(tile, st):
            def f(i):
                ops.gemm_raw(As[i], K, w4, 9 * K, Cs[i], N, M, N, K, taps=9, bias=bias, relu=True, geom=g, B_split=sp, precision=1, A16=Ah[i], A16lo=Al[i],
                             C16=C16[i], dl=(tile, st))
            return f
    else:
        def mk(tile, st):   # inner data gradient of a bottleneck: twin in, ReLU gate from the twin, twin-only out
            def f(i):
                ops.gemm_raw(None, K, w4, 9 * K, None, N, M, N, K, taps=9, geom=g, B_split=sp, B16=w16, precision=3, A16=Ah[i], C16=C16[i],
                             gate=Cs[i], ldg=N, gate16=Gt[i], dl=(tile, st))
            return f
        row["old"] = bench(mk(0 if (9 * K >= 2048 and N % 128 == 0 and (M // 128) * (N // 128) >= 128) else 3, 3), nsets)
    row["dl64"] = bench(mk(3, 3), nsets)
    for t in TILES:
        try:
            row[t] = bench(mk(t, 13), nsets)
        except RuntimeError as e:
            row[t] = float("nan")
    return row, 2.0 * M * N * K * 9


def main():
    mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for precision, name, col in ((1, "forward, split-bf16 x3 (old = register-staged on fp32 rows)", 5), (3, "data gradient, plain bf16 (old = classic direct-to-LDS, product tile)", 6)):
        print(f"== {name}; us per launch (TF algorithmic); images x{mult}")
        print("%-26s %14s %14s | " % ("N H W C dil", "old", "classic 64x64") + " ".join("%14s" % ("halo " + TILES[t]) for t in TILES))
        tot_old = tot_best = 0.0
        for sh in SHAPES:
            if sh[col] == 0:
                continue
            row, fl = run(sh, precision, mult)
            cell = lambda us: "%6.1f (%4.0f)" % (us, fl / us / 1e6)      # noqa: E731
            ok = [(row[t], t) for t in TILES if row[t] == row[t]]
            if not ok:
                print("%-26s %14s %14s | halo: not eligible (more than 448 halo rows)" % (str(sh[:5]), cell(row["old"]), cell(row["dl64"])))
                continue
            best = min(ok)
            print("%-26s %14s %14s | " % (str(sh[:5]), cell(row["old"]), cell(row["dl64"])) + " ".join("%14s" % cell(row[t]) for t in TILES) +
                  "   best halo %s x%.2f vs old" % (TILES[best[1]], row["old"] / best[0]))
            tot_old += sh[col] * row["old"]
            tot_best += sh[col] * min(best[0], row["old"])
        print("per step (calls weighted): old %.0f us, best-of(old, halo) %.0f us" % (tot_old, tot_best))


if __name__ == "__main__":
    main()
