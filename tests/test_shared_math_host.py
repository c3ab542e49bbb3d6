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
            os.path.join(O.ROOT, "2dgaussiansplatting_amd", "csrc", "s2d_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                               "-o", so, srcs[0], "-lm"])
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
