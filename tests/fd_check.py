"""End-to-end finite-difference check of the analytic backward pass.

The reference validates its hand-derived gradients with commented finite-difference prints next to each formula
(main.cpp:642-701) and a switch that makes exp_approx the true exponential (main.cpp:51), because the formulas are the
derivatives of exp, not of (1 + x/8)^8.  This module does that validation end to end, for any implementation of the
forward and backward pass that honours the switch: central differences of the loss the backward pass differentiates,

    L(splats) = 1/2 * sum over pixels and channels of (image0 - imageRef)^2        (dL/dC = C - ref, main.cpp:616)

through the real forward rasteriser, against all nine gradient components of every splat.

The forward pass is only piecewise smooth: the footprint is a set of whole pixels (int truncation, main.cpp:490-491,
:507-508) and the early-out at T < 1/256 (:520).  A parameter step that flips a pixel in or out moves L by a jump that
does not scale with the step, so each derivative is taken at two step sizes (h, 2h); pairs whose two estimates
disagree are discontinuous on that interval and are not used (their count is reported and bounded).
"""
import numpy as np

# steps per parameter (pos.x, pos.y, sx, sy, rot, r, g, b, opacity): small against the 1-pixel footprint granularity,
# large against the fp32 noise of the forward pass (see loss_delta)
STEPS = np.array([2e-3, 2e-3, 2e-3, 2e-3, 2e-4, 1e-2, 1e-2, 1e-2, 1e-2])
NAMES = ["pos.x", "pos.y", "sx", "sy", "rot", "color.r", "color.g", "color.b", "opacity"]


def loss_delta(img_plus, img_minus, ref):
    """L(+) - L(-) in double, pixel by pixel: pixels the perturbed splat does not reach are bit-identical in the two
    images and cancel exactly, so the fp32 rounding of the untouched 99 % of the image never enters."""
    p = img_plus[..., :3].astype(np.float64)
    m = img_minus[..., :3].astype(np.float64)
    r = ref[..., :3].astype(np.float64)
    return 0.5 * float((((p - r) ** 2) - ((m - r) ** 2)).sum())


def central_difference(render, splats9, i, k, h, ref):
    """d L / d splats9[i, k] by a central difference of half-width ~h (the actual binary32 step is used)."""
    base = np.float32(splats9[i, k])
    hi, lo = np.float32(base + np.float32(h)), np.float32(base - np.float32(h))
    s = splats9.copy()
    s[i, k] = hi
    ip = render(s).copy()
    s[i, k] = lo
    im = render(s).copy()
    return loss_delta(ip, im, ref) / (float(hi) - float(lo))


def check(render, analytic, splats9, ref, which=None, rtol=1e-2, consist=4e-3):
    """render(splats9) -> image0 (H, W, 4) float32; analytic: (n, 9) gradients of L at splats9.
    Returns a dict of statistics; raises AssertionError when the analytic gradients are not the derivative."""
    n = splats9.shape[0]
    which = range(n) if which is None else which
    scale = np.maximum(np.median(np.abs(analytic), axis=0), 1e-12)   # typical magnitude per parameter kind
    used, skipped, worst, worst_at = 0, 0, 0.0, None
    per_param = np.zeros(9)
    for i in which:
        for k in range(9):
            d1 = central_difference(render, splats9, i, k, STEPS[k], ref)
            d2 = central_difference(render, splats9, i, k, 2.0 * STEPS[k], ref)
            g = float(analytic[i, k])
            floor = 0.05 * scale[k]
            if abs(d1 - d2) > consist * max(abs(d1), abs(d2), floor):
                skipped += 1      # a footprint pixel or an early-out flipped inside the interval
                continue
            err = abs(g - d1) / max(abs(g), abs(d1), floor)
            used += 1
            per_param[k] = max(per_param[k], err)
            if err > worst:
                worst, worst_at = err, (i, NAMES[k], g, d1)
    stats = {"used": used, "skipped": skipped, "worst_rel": worst, "worst_at": worst_at,
             "per_param": dict(zip(NAMES, per_param.tolist()))}
    assert used >= 0.6 * (used + skipped) and used >= 40, stats
    assert worst <= rtol, stats
    return stats


def scene(W=72, H=56, n=14, seed=3):
    """A small scene with every regime the formulas see: overlapping anisotropic splats, partial opacity, a few
    splats crossing the image border, a smooth coloured target."""
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 9), dtype=np.float32)
    s[:, 0] = rng.uniform(4, W - 4, n)
    s[:, 1] = rng.uniform(4, H - 4, n)
    s[:, 2] = rng.uniform(2.5, 7.0, n)
    s[:, 3] = rng.uniform(2.5, 7.0, n)
    s[:, 4] = rng.uniform(0, np.pi, n)
    s[:, 5:8] = rng.uniform(0.1, 0.9, (n, 3))
    s[:, 8] = rng.uniform(0.25, 0.9, n)
    s[0, 0], s[0, 1] = 1.5, 2.5          # hangs over the top-left corner
    s[1, 0] = W - 2.0                    # over the right edge
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    ref = np.empty((H, W, 4), dtype=np.float32)
    ref[..., 0] = 0.5 + 0.4 * np.sin(xx / 9.0)
    ref[..., 1] = 0.5 + 0.4 * np.cos(yy / 7.0)
    ref[..., 2] = (xx + yy) / (W + H)
    ref[..., 3] = 1.0
    return s, ref
