"""The reference's own (commented-out) "numerical varidation" blocks, run: main.cpp:642-652 (d alpha / d v),
:664-675 (d alpha / d sx, sy), :687-701 (d alpha / d theta), plus Form.pdf sections 6-7 (d C / d c_i,
d C / d alpha_i with the colour-from-behind term S_i).  Everything in float64 with the true exponential, as the
reference prescribes for these checks (main.cpp:51: "return expf( x ); // use this for numerical varidation").
They validate the analytic formulas that oracle/s2d_oracle.c and csrc/s2d_raster.hip both implement
(SURVEY.md section 8 row f4); they do not involve the GPU.
"""
import numpy as np
import pytest

RNG = np.random.default_rng(2024)


def inv_cov(sx, sy, th):
    c, s = np.cos(th), np.sin(th)
    l0, l1 = sx * sx, sy * sy
    s11 = l0 * c * c + l1 * s * s          # cov_of, main.cpp:206-221
    s12 = (l0 - l1) * s * c
    s22 = l0 + l1 - s11
    det = s11 * s22 - s12 * s12
    return s22 / det, -s12 / det, s11 / det  # a, b (= c), d; main.cpp:432-436


def alpha(vx, vy, sx, sy, th, o):
    a, b, d = inv_cov(sx, sy, th)
    d2 = vx * (a * vx + b * vy) + vy * (b * vx + d * vy)   # main.cpp:526
    return o * np.exp(-0.5 * d2)                            # main.cpp:527 with expf


def analytic(vx, vy, sx, sy, th, o):
    """The expressions of main.cpp:639-640, :657-662, :680-683, :703, verbatim in float64."""
    a, b, d = inv_cov(sx, sy, th)
    c_ = b
    al = alpha(vx, vy, sx, sy, th, o)
    cos, sin = np.cos(th), np.sin(th)
    dalpha_dx = 0.5 * al * (2.0 * a * vx + (b + c_) * vy)
    dalpha_dy = 0.5 * al * (2.0 * d * vy + (b + c_) * vx)
    dot3 = lambda p, q: p[0] * q[0] + p[1] * q[1] + p[2] * q[2]
    vv = (vx * vx, vx * vy, vy * vy)
    dalpha_dsx = al / (sx * sx * sx) * dot3((cos * cos, 2.0 * sin * cos, sin * sin), vv)
    dalpha_dsy = al / (sy * sy * sy) * dot3((sin * sin, -2.0 * sin * cos, cos * cos), vv)
    dalpha_dth = al * (sx * sx - sy * sy) / (sx * sx * sy * sy) * ((cos * cos - sin * sin) * vx * vy - sin * cos * (vx * vx - vy * vy))
    dalpha_do = al / o
    return dalpha_dx, dalpha_dy, dalpha_dsx, dalpha_dsy, dalpha_dth, dalpha_do


@pytest.mark.parametrize("trial", range(40))
def test_alpha_derivatives_match_central_differences(trial):
    vx, vy = RNG.uniform(-12, 12, 2)
    sx, sy = RNG.uniform(1.5, 12, 2)
    th = RNG.uniform(-4, 4)
    o = RNG.uniform(0.1, 1.0)
    dx, dy, dsx, dsy, dth, do = analytic(vx, vy, sx, sy, th, o)
    h = 1e-6
    fd = lambda f: (f(+h) - f(-h)) / (2 * h)
    # the position derivative is with respect to mu = pixel - v, hence the sign (main.cpp:642: "this is just for v not mu")
    tol = dict(rel=2e-6, abs=1e-9)
    assert -fd(lambda e: alpha(vx + e, vy, sx, sy, th, o)) == pytest.approx(dx, **tol)
    assert -fd(lambda e: alpha(vx, vy + e, sx, sy, th, o)) == pytest.approx(dy, **tol)
    assert fd(lambda e: alpha(vx, vy, sx + e, sy, th, o)) == pytest.approx(dsx, **tol)
    assert fd(lambda e: alpha(vx, vy, sx, sy + e, th, o)) == pytest.approx(dsy, **tol)
    assert fd(lambda e: alpha(vx, vy, sx, sy, th + e, o)) == pytest.approx(dth, **tol)
    assert fd(lambda e: alpha(vx, vy, sx, sy, th, o + e)) == pytest.approx(do, **tol)


def blend(colors, alphas):
    """C = sum_i c_i alpha_i prod_{j<i} (1 - alpha_j), Form.pdf section 5 / main.cpp:529-533."""
    T, C = 1.0, np.zeros(3)
    for c, a in zip(colors, alphas):
        C = C + T * c * a
        T = T * (1.0 - a)
    return C


@pytest.mark.parametrize("trial", range(10))
def test_colour_and_alpha_derivatives_of_the_blend(trial):
    n = 7
    colors = RNG.uniform(0, 1, (n, 3))
    alphas = RNG.uniform(0.02, 0.9, n)
    ref = RNG.uniform(0, 1, 3)
    C = blend(colors, alphas)
    dL_dC = C - ref                                   # main.cpp:616 (loss = 1/2 |C - ref|^2)
    loss = lambda cs, als: 0.5 * np.sum((blend(cs, als) - ref) ** 2)
    T = 1.0
    run = np.zeros(3)
    h = 1e-6
    for i in range(n):
        # dL/dc_i = dL/dC * alpha_i * T_i, main.cpp:618-619
        for k in range(3):
            cp, cm = colors.copy(), colors.copy()
            cp[i, k] += h
            cm[i, k] -= h
            assert (loss(cp, alphas) - loss(cm, alphas)) / (2 * h) == pytest.approx(dL_dC[k] * alphas[i] * T, rel=1e-6, abs=1e-10)
        run = run + T * colors[i] * alphas[i]            # main.cpp:623-625
        S = C - run                                      # colour from behind, main.cpp:627
        dC_dalpha = colors[i] * T - S / (1.0 - alphas[i])  # main.cpp:628
        ap, am = alphas.copy(), alphas.copy()
        ap[i] += h
        am[i] -= h
        assert (loss(colors, ap) - loss(colors, am)) / (2 * h) == pytest.approx(float(np.dot(dL_dC, dC_dalpha)), rel=1e-6, abs=1e-10)
        T = T * (1.0 - alphas[i])                        # main.cpp:707


def test_regrouped_forms_used_by_the_kernel_are_identities():
    """s2d_raster.hip evaluates 0.5*alpha*(2a vx + (b+c) vy) as alpha*(a vx + b vy); DESIGN.md section 5 also quotes
    the u/w identities.  Check them as algebra."""
    for _ in range(100):
        vx, vy = RNG.uniform(-20, 20, 2)
        sx, sy = RNG.uniform(1, 20, 2)
        th = RNG.uniform(-4, 4)
        a, b, d = inv_cov(sx, sy, th)
        c, s = np.cos(th), np.sin(th)
        assert 0.5 * (2 * a * vx + (b + b) * vy) == pytest.approx(a * vx + b * vy, rel=1e-12)
        u, w = c * vx + s * vy, s * vx - c * vy
        assert c * c * vx * vx + 2 * s * c * vx * vy + s * s * vy * vy == pytest.approx(u * u, rel=1e-9, abs=1e-9)
        assert s * s * vx * vx - 2 * s * c * vx * vy + c * c * vy * vy == pytest.approx(w * w, rel=1e-9, abs=1e-9)
        assert (c * c - s * s) * vx * vy - s * c * (vx * vx - vy * vy) == pytest.approx(-u * w, rel=1e-9, abs=1e-9)
