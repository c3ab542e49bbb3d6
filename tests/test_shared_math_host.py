"""Unit tests (CPU) of the arithmetic header the HIP kernels are built from, against the oracle.

tests/hostcheck/s2d_hostcheck.cpp compiles 2dgaussiansplatting_amd/csrc/s2d_math.h for the host.
These tests do not exercise the product path (that needs a GPU: see test_gpu_*.py); they catch
restatement errors in the shared math before any GPU time is spent.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
HC_DIR = os.path.join(HERE, "hostcheck")


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HC_DIR, "libs2d_hostcheck.so")
    srcs = [os.path.join(HC_DIR, "s2d_hostcheck.cpp"),
            os.path.join(O.ROOT, "2dgaussiansplatting_amd", "csrc", "s2d_math.h"),
            os.path.join(O.ROOT, "2dgaussiansplatting_amd", "host", "overlay.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-I", os.path.join(O.ROOT, "include"),
                               "-o", so, srcs[0], "-lm", "-lz"])
    L = C.CDLL(so)
    L.hc_adam.restype = C.c_float
    L.hc_adam.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 5
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


# the 34 arguments (|x| < 120) where an un-fused evaluation of the same polynomials differs from libm
SENSITIVE = np.array([0x4255b0a9, 0x418a3adb, 0xc255b0a9, 0xc18a3adb], dtype=np.uint32).view(np.float32)


def test_trig_matches_libm_bitwise(hc):
    rng = np.random.default_rng(7)
    xs = np.concatenate([
        rng.uniform(-119.9, 119.9, 2_000_000).astype(np.float32),
        rng.uniform(-4, 4, 1_000_000).astype(np.float32),
        (rng.standard_normal(200_000) * 1e-3).astype(np.float32),
        np.array([0.0, -0.0, 1e-30, 2.0 ** -12, np.pi / 4, np.pi / 2, np.pi, 3.1415927, 119.99999], dtype=np.float32),
        SENSITIVE])
    s = np.empty_like(xs)
    c = np.empty_like(xs)
    hc.hc_sincos(p(xs), len(xs), p(s), p(c))
    L = O.lib()
    # oracle's libm, vectorised through numpy would use a different implementation: call the oracle itself
    idx = rng.choice(len(xs), 300_000, replace=False)
    idx = np.concatenate([idx, np.arange(len(xs) - 13, len(xs))])
    for i in idx:
        x = float(xs[i])
        assert np.float32(L.s2do_sinf(x)).view(np.uint32) == s[i].view(np.uint32), x
        assert np.float32(L.s2do_cosf(x)).view(np.uint32) == c[i].view(np.uint32), x


def test_init_bitwise(hc):
    for (W, H, n) in ((268, 213, 2000), (535, 426, 5000), (4096, 4096, 10000)):
        a = np.zeros(n, dtype=O.SPLAT_DTYPE)
        hc.hc_init(p(a), n, W, H)
        b = np.zeros(n, dtype=O.SPLAT_DTYPE)
        O.lib().s2do_init(p(b), None, n, W, H)
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("n,steps", [(1024, 0), (2000, 3)])
def test_forward_rule_bitwise(hc, n, steps):
    """row_mask16 + gauss_at + blend order reproduce the oracle's framebuffer bit for bit."""
    tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
    t = O.OracleTrainer(tgt, n)
    for _ in range(steps):
        t.step()
    want = t.forward().copy()
    got = np.zeros_like(want)
    hc.hc_forward(p(t.splats), n, t.W, t.H, p(got))
    assert got.tobytes() == want.tobytes()


def test_forward_rule_bitwise_anisotropic(hc):
    """Hand-made splats: thin, huge, rotated, partly/fully off-image, low opacity."""
    W, H = 96, 80
    rng = np.random.default_rng(3)
    n = 300
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    s["pos"][:, 0] = rng.uniform(0, W - 1, n)
    s["pos"][:, 1] = rng.uniform(0, H - 1, n)
    s["sx"] = rng.choice([1.0, 1.5, 3.0, 8.0, 40.0, 300.0, 1024.0], n)
    s["sy"] = rng.choice([1.0, 2.0, 6.0, 25.0, 1024.0], n)
    s["rot"] = rng.uniform(-7, 7, n)
    s["color"] = rng.uniform(0, 1, (n, 3))
    s["opacity"] = rng.uniform(0.1, 1.0, n)
    s["pos"][:10] = [[0, 0], [W - 1, H - 1], [0, H - 1], [W - 1, 0], [47.5, 39.5], [0.5, 0.5], [16, 16], [15.999, 31.999], [32, 48], [95, 79]]
    tgt = O.synthetic_target(W, H)
    t = O.OracleTrainer(tgt, n)
    t.splats[:] = s
    want = t.forward().copy()
    got = np.zeros_like(want)
    hc.hc_forward(p(t.splats), n, W, H, p(got))
    assert got.tobytes() == want.tobytes()
    assert hc.hc_check_bounds(p(t.splats), n, W, H) == 0


def test_binning_bounds_contain_exact_ranges(hc):
    tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
    t = O.OracleTrainer(tgt, 2000)
    for _ in range(5):
        t.step()
    assert hc.hc_check_bounds(p(t.splats), t.n, t.W, t.H) == 0


def test_adam_scalar_bitwise(hc):
    rng = np.random.default_rng(11)
    L = O.lib()
    n = 512
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    L.s2do_init(p(s), None, n, 300, 200)
    a = np.zeros(n, dtype=O.ADAM_DTYPE)
    g = np.zeros(n, dtype=O.SPLAT_DTYPE)
    gv = g.view(np.float32).reshape(n, 9)
    b1 = np.ones(1, dtype=np.float32)
    b2 = np.ones(1, dtype=np.float32)
    s2 = s.copy()
    a2 = a.copy()
    for it in range(4):
        gv[:] = (rng.standard_normal((n, 9)) * 10.0 ** rng.uniform(-6, 2, (n, 9))).astype(np.float32)
        b1_prev, b2_prev = b1.copy(), b2.copy()
        assert L.s2do_adam_step(p(s), p(a), p(g), n, 300, 200, p(b1), p(b2), 1, 0.05) == 0
        # same through the shared helper, scalar by scalar (order: pos[2], sx, sy, rot, color[3], opacity)
        sv = s2.view(np.float32).reshape(n, 9)
        av = a2.view(np.float32).reshape(n, 9, 2)
        nb1 = np.float32(b1_prev[0] * np.float32(0.9))
        nb2 = np.float32(b2_prev[0] * np.float32(0.99))
        assert nb1 == b1[0] and nb2 == b2[0]
        for i in range(0, n, 7):
            for k in range(9):
                m = C.c_float(av[i, k, 0])
                v = C.c_float(av[i, k, 1])
                r = hc.hc_adam(C.byref(m), C.byref(v), float(sv[i, k]), float(gv[i, k]), 0.05, float(nb1), float(nb2))
                lo, hi = {0: (0, 299), 1: (0, 199), 2: (1, 1024), 3: (1, 1024), 4: (None, None),
                          5: (0, 1), 6: (0, 1), 7: (0, 1), 8: (0.1, 1)}[k]
                r = np.float32(r)
                if lo is not None:
                    r = np.float32(min(max(r, np.float32(lo)), np.float32(hi)))
                assert r == s.view(np.float32).reshape(n, 9)[i, k], (it, i, k)
                assert np.float32(m.value) == a.view(np.float32).reshape(n, 9, 2)[i, k, 0]
                assert np.float32(v.value) == a.view(np.float32).reshape(n, 9, 2)[i, k, 1]
        s2[:] = s
        a2[:] = a


# ---------------------------------------------------------------------------------------------
# row f3: the splat overlay's vertices (host/overlay.h) against the oracle's restatement of main.cpp:419-477
# ---------------------------------------------------------------------------------------------
def _overlay_cases():
    o = O.OracleTrainer(O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di"))), 1024)
    init = o.splats.copy()
    for _ in range(10):
        o.step()
    trained = o.splats.copy()
    rng = np.random.default_rng(7)
    odd = np.zeros(64, dtype=O.SPLAT_DTYPE)
    odd["pos"] = rng.uniform(-20, 300, (64, 2))
    odd["sx"] = rng.choice([1.0, 1.0000001, 3.5, 1024.0, 7.25], 64)
    odd["sy"] = rng.choice([1.0, 2.0, 3.5, 1024.0, 7.25], 64)           # sx == sy: s12 = 0, the eps of main.cpp:229 decides
    odd["rot"] = rng.choice([0.0, 1e-8, np.pi / 2, -3.0, 100.0, 0.7853982], 64)
    odd["color"] = rng.uniform(0, 1, (64, 3))
    odd["opacity"] = 1.0
    return [("init", init), ("after 10 iterations", trained), ("degenerate", odd)]


def test_overlay_vertices_equal_the_reference_restatement_bit_for_bit(hc):
    """The 46 PrimVertex calls per splat of main.cpp:447-476 -- the two axes pos +- eigen * sqrt(lambda) (:443-451), the
    16-gon (:454-462) and the 1-sigma box (:464-477) -- as the host program draws them (host/overlay.h), against the
    oracle's restatement (oracle/s2d_oracle.c): floats bitwise, colours as the u8 triples.  Fails if an axis end point,
    an ellipse vertex or a box corner moves by one ulp."""
    hc.hc_overlay_vertices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for name, splats in _overlay_cases():
        want_xyz, want_rgb = O.overlay_vertices(splats)
        sp = np.ascontiguousarray(splats).view(np.float32).reshape(-1, 9)
        xyz = np.zeros_like(want_xyz)
        rgb = np.zeros_like(want_rgb)
        hc.hc_overlay_vertices(p(sp), sp.shape[0], p(xyz), p(rgb))
        assert xyz.view(np.uint32).tolist() == want_xyz.view(np.uint32).tolist(), name
        assert rgb.tobytes() == want_rgb.tobytes(), name


def test_overlay_vertices_have_the_reference_structure():
    """What the restatement must reproduce of main.cpp:441-477, checked on the oracle's list itself: vertex order and
    colours (:447-451, :473-474), the closed 17-segment 16-gon around pos whose first vertex is pos + axis1 (sin 0 = 0,
    cos 0 = 1, :459), orthogonal axes of length sqrt(lambda), the box half sizes sqrt(s11), sqrt(s22) (Form.pdf section 12)."""
    for name, splats in _overlay_cases()[:2]:
        xyz, rgb = O.overlay_vertices(splats)
        n = len(splats)
        v = xyz.reshape(n, 46, 3).astype(np.float64)
        c = rgb.reshape(n, 46, 3)
        sp = splats.view(np.float32).reshape(-1, 9).astype(np.float64)
        pos = np.stack([sp[:, 0], -sp[:, 1]], axis=1)
        assert (v[:, :, 2] == 0).all()
        assert (v[:, 0, :2] == pos).all() and (v[:, 2, :2] == pos).all()
        assert (c[:, 0:3] == 255).all() and (c[:, 3] == 230).all() and (c[:, 38:] == 128).all()
        assert (c[:, 4:38] == (splats["color"] * np.float32(255.0)).astype(np.uint32).astype(np.uint8)[:, None, :]).all()
        a0, a1 = v[:, 1, :2] - pos, v[:, 3, :2] - pos
        l0, l1 = np.maximum(sp[:, 2], sp[:, 3]) ** 2, np.minimum(sp[:, 2], sp[:, 3]) ** 2
        np.testing.assert_allclose((a0 ** 2).sum(1), l0, rtol=2e-3)     # |axis0|^2 = lambda0 (the larger), main.cpp:443
        np.testing.assert_allclose((a1 ** 2).sum(1), l1, rtol=2e-3, atol=2e-3)
        assert (np.abs((a0 * a1).sum(1)) <= 1e-5 * np.sqrt(l0 * l1)).all()
        ell = v[:, 4:38, :2]
        np.testing.assert_array_equal(ell[:, 1:-1:2], ell[:, 2::2])     # a polyline: each segment starts where the last ended
        np.testing.assert_allclose(ell[:, 0] - pos, a1, atol=1e-4)      # circular.sin() = 0, .cos() = 1 at the start
        np.testing.assert_allclose(ell[:, 32], ell[:, 0], atol=2e-3)    # closed after 16 steps (the 17th segment repeats the first)
        # the box: corners (-hx,-hy) (hx,-hy) (hx,hy) (-hx,hy), hx^2 = s11, hy^2 = s22 (cov_of, main.cpp:206-221)
        ct, st = np.cos(sp[:, 4]), np.sin(sp[:, 4])
        s11 = sp[:, 2] ** 2 * ct * ct + sp[:, 3] ** 2 * st * st
        s22 = sp[:, 2] ** 2 + sp[:, 3] ** 2 - s11
        box = v[:, 38:, :2] - pos[:, None, :]
        np.testing.assert_allclose(box[:, 0], np.stack([-np.sqrt(s11), -np.sqrt(s22)], 1), rtol=1e-4)
        np.testing.assert_allclose(box[:, 3], np.stack([np.sqrt(s11), np.sqrt(s22)], 1), rtol=1e-4)
        np.testing.assert_array_equal(box[:, 1:-1:2], box[:, 2::2])
        np.testing.assert_array_equal(box[:, 7], box[:, 0])
