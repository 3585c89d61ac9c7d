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


def grads_well_posed(z, name, floor=1e-4):
    """Element-wise GRADIENT bars are a well-posed demand only when no matched box coordinate sits on the kink of the L1 loss: the generator
    stores the smallest |pred - target| over the matched coordinates (`min_l1_margin`).  cfg2 (bench.py's batch) has 1.8e-7 -- below one
    fp32 ulp: the sign of that coordinate's gradient is a coin toss for ANY arithmetic, and the flip moves every parameter's gradient by
    up to ~1 % (measured: 14 % on the box head's cy-bias, whose 157 terms nearly cancel); its norms / sums / post-AdamW signs stay checked.
    lvis_wide (2.6e-4) carries the element-wise gradient bars at full size."""
    return bool(z[f"{name}/min_l1_margin"].min() >= floor)


def check_param_samples(z, name, model, grads=True, grad_rtol=3e-2, what="", post=True):
    """Element-wise bars on top of the norm / sum budgets: the generator stored, for every parameter with a gradient, its
    SAMPLE_K largest-|gradient| elements -- flat index, raw gradient, value before and after the reference's clip + AdamW step.
      * gradient (grads=True: p.grad still holds this step's raw gradient): every sampled element within grad_rtol of the reference's
        (plain-bf16 backward: measured ~3e-3; the bar is 3e-2, ten times tighter than a sign flip and 3x the per-parameter norm bar);
      * post-step value: the first AdamW update moves an element by lr * g / (|g| + eps) ~ lr * sign(g) plus the decay: the sampled
        elements must land within 0.05 * lr of the reference's value (a flipped sign is 2 * lr away).
    -> (worst relative gradient error, worst post-step error in units of lr); both are printed by the callers so that a 10x regression
    inside the bars is visible in the log."""
    names = [str(n) for n in z[f"{name}/param_names"]]
    params = dict(model.named_parameters())
    pidx, fidx = z[f"{name}/sample_pidx"], z[f"{name}/sample_fidx"]
    g_ref, after = z[f"{name}/sample_grad"], z[f"{name}/sample_after"]
    worst_g, worst_p, n_rel = 0.0, 0.0, 0
    tn = float(z[f"{name}/grad_total_norm"])
    by_param = {}
    for k in range(len(pidx)):
        by_param.setdefault(int(pidx[k]), []).append(k)
    for pi, ks in by_param.items():
        n = names[pi]
        p = params[n]
        lr = 1e-5 if "backbone" in n else 1e-4
        fi = torch.as_tensor(fidx[ks], device=p.device)
        if post:                 # (post=False: the caller ran a backward without an optimizer step -- gradients only)
            vals = p.detach().reshape(-1)[fi].double().cpu().numpy()
            err_p = np.abs(vals - after[ks].astype(np.float64)) / lr
            assert err_p.max() <= 0.05, f"{what}{n}: post-AdamW element {int(fidx[ks][err_p.argmax()])} is {err_p.max():.3f} lr from the reference's"
            worst_p = max(worst_p, float(err_p.max()))
        if grads and grads_well_posed(z, name):
            g = p.grad.reshape(-1)[fi].double().cpu().numpy()
            gr = g_ref[ks].astype(np.float64)
            # relative bar for the elements that matter to the step (|g| >= 1e-5 of the total gradient norm); below that a parameter's whole
            # gradient is a sum that nearly cancels (adapt_pos2d.2.bias in lvis_wide: 3e-6 against a total norm of 428 -- the plain-bf16
            # backward's 2^-9 per term shows as 6 % there): those get the absolute floor the per-parameter norm bar uses (1e-6 of the total norm)
            big = np.abs(gr) >= 1e-5 * tn
            err_abs = np.abs(g - gr)
            assert (err_abs[~big] <= 1e-6 * tn).all(), f"{what}{n}: small gradient element off by {err_abs[~big].max():.3e} (total norm {tn:.3e})"
            if big.any():
                err_g = err_abs[big] / np.abs(gr[big])
                k = int(np.flatnonzero(big)[err_g.argmax()])
                assert err_g.max() <= grad_rtol, f"{what}{n}: gradient element {int(fidx[ks][k])}: {g[k]:.6e} vs {gr[k]:.6e}"
                worst_g = max(worst_g, float(err_g.max()))
                n_rel += int(big.sum())
    check_param_samples.last_counts = (n_rel, len(pidx))      # (elements under the relative gradient bar, elements sampled)
    return worst_g, worst_p
