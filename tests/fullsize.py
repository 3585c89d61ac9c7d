"""Shared helpers of the full-size parity tests (tests/golden/g10_full.npz, g11_stage1_n900.npz): the inputs are regenerated
from the recorded seeds (oracle.step), intermediates are compared through the strided digests the generator stored."""
import numpy as np
import torch

from oracle.step import synthetic_batch, synthetic_images

NSAMP, FULL_MAX = 2048, 4096          # oracle/gen_golden_full.py: G.digest(v, full_max=4096, nsamp=2048)


def case_inputs(z, name):
    B, nq, is_grid, seed, aux, nq_eff = [int(v) for v in z[f"{name}/cfg"]]
    sizes = [tuple(int(x) for x in s) for s in z[f"{name}/sizes"]]
    Ts = tuple(int(t) for t in z[f"{name}/Ts"])
    if len(set(sizes)) == 1:
        images, rects, targets = synthetic_batch(B=B, H=sizes[0][0], W=sizes[0][1], Ts=Ts, seed=seed)
    else:
        images, rects, targets = synthetic_images(sizes, Ts, seed)
    assert np.array_equal(rects.numpy(), z[f"{name}/rects"])
    for b, t in enumerate(targets):
        assert np.array_equal(t["boxes"].numpy(), z[f"{name}/tgt{b}"]), "seeded inputs differ from the generator's"
    return dict(B=B, nq=nq, prior="grid" if is_grid else "learned", aux=bool(aux), nq_eff=nq_eff, images=images, rects=rects,
                targets=targets)


def rel_err(a, b):
    """max |a - b| / max |b| -- the 1e-3 'relative' bar of north_star, taken against the tensor's scale."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elementwise_rel_err(a, b, floor=1e-2):
    """north_star's "within 1e-3 rel" read ELEMENT-WISE: max over elements of |a - b| / |b|, for the elements whose reference magnitude
    is at least `floor` x the tensor's scale (below that a relative error is a statement about rounding noise around zero; those
    elements are held to floor x 1e-3 x scale absolutely by rel_err).  -> (worst relative error, share of elements it covers)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    big = np.abs(b) >= floor * max(np.abs(b).max(), 1e-30)
    if not big.any():
        return 0.0, 0.0
    return float((np.abs(a - b)[big] / np.abs(b)[big]).max()), float(big.mean())


def check_tap(z, key, t, tol, what=""):
    """t: tensor in the REFERENCE's layout (NCHW feature maps, [B,Q,C] decoder states)."""
    t = t.detach().to(torch.float64).reshape(-1).cpu()
    if f"{key}/full" in z:
        e = rel_err(t.numpy(), z[f"{key}/full"])
        assert e <= tol, f"{what or key}: {e:.2e} > {tol:.0e}"
        return e
    step = int(z[f"{key}/step"])
    e = rel_err(t[::step][:NSAMP].numpy(), z[f"{key}/sample"])
    assert e <= tol, f"{what or key}: sample {e:.2e} > {tol:.0e}"
    stats = z[f"{key}/stats"]
    np.testing.assert_allclose(t.norm().item(), stats[0], rtol=tol, err_msg=f"{key} l2")
    np.testing.assert_allclose(t.abs().sum().item(), stats[2], rtol=tol, err_msg=f"{key} abs-sum")
    return e


TAP_KEYS = ["layer4", "proj"] + [f"enc{i}" for i in range(6)] + [f"hs{i}" for i in range(6)]
