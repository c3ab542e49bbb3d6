"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (north_star: "within 1e-4 rel fp32"):
  * integer / index work (init hash, scan, sort, tile lists, pair counts): bit-exact;
  * forward framebuffer: BIT-EXACT (the kernels keep the reference's operation order, no FMA
    contraction, correctly rounded div/sqrt, and a sinf/cosf that matches the oracle's libm bit for bit);
  * gradients: the GPU adds up the same fp32 per-pixel contributions as the reference, in a different order
    (per wave by DPP, per tile in LDS, across tiles by float atomics; the reference: sequentially, row-major).
    The oracle therefore also returns dsum (those contributions summed in double) and dabs (the sum of their
    magnitudes), and three things are asserted for every (splat, component):
      (a) |gpu - dsum| <= 1e-6 * dabs          the GPU sum is the exact sum to fp32 summation accuracy,
      (b) max|gpu - dsum|/dabs <= max|oracle - dsum|/dabs   ... and at least as close to it as the reference's
                                                own sequential fp32 sum (measured: ~3e-7 vs ~2.5e-6),
      (c) |gpu - oracle| <= 1e-4 * max(|oracle|, 0.02 * dabs)   1e-4 relative wherever the sum keeps >= 2 % of
                                                its terms' magnitude; sums that cancel harder than 50:1 are
                                                held to 2e-6 of the term magnitude instead (measured max 5e-5);
      The literal "1e-4 of the result" is also MEASURED and reported (not asserted) for every case -- the share of gradient
      scalars within 1e-4 of the oracle's value, next to the share of the ORACLE's own fp32 sums within 1e-4 of the exact
      sum of its terms: sums that cancel cannot be held to 1e-4 of themselves by any fp32 summation order, the
      reference's included (profiles/r03/r03_parity_report.txt);
  * one optimiser step from identical state, judged on the UPDATE: |delta_gpu - delta_oracle| <= 1e-4 of lr per scalar
    beyond one ulp of the parameter (oracle_lib.step_delta_error), and the parameters themselves within 1e-4 relative
    (floor 1.0) of the oracle's;
  * MSE (double): 1e-9 relative for the same framebuffer.
"""
import hashlib
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

S2D = importlib.import_module("2dgaussiansplatting_amd")
MINI = os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")
FULL = os.path.join(O.GOLDEN, "squirrel_cls_535x426.s2di")
REL = 1e-4
STEP_REL = 1e-4   # |delta_gpu - delta_oracle| <= 1e-4 of lr per scalar, beyond one ulp of the parameter (measured <= 1e-5)


def mini_target():
    return O.target_rgba32f(O.load_s2di(MINI))


def make_pair(target, n, steps=0, opacity=False, **kw):
    """Oracle advanced `steps` iterations, and a GPU trainer loaded with the oracle's state."""
    o = O.OracleTrainer(target, n, optimize_opacity=opacity)
    for _ in range(steps):
        o.step()
    t = S2D.Trainer(o.W, o.H, n, **kw)
    t.set_target(target)
    t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
    t.set_adam(o.adams.view(S2D.ADAM_DTYPE), o.beta1t[0], o.beta2t[0], o.iterations)
    t.optimize_opacity = opacity
    return o, t


def grad_check(got, oracle):
    """Asserts the three gradient bars of the module docstring; `oracle` has just run forward()."""
    w32, dsum, dabs = oracle.backward_stats()
    return O.grad_bars(got.view(np.float32), w32.view(np.float32), dsum, dabs, REL)["c_gpu_vs_oracle"]


def random_splats(n, W, H, seed):
    rng = np.random.default_rng(seed)
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    s["pos"][:, 0] = rng.uniform(0, W - 1, n)
    s["pos"][:, 1] = rng.uniform(0, H - 1, n)
    s["sx"] = rng.choice([1.0, 1.5, 3.0, 8.0, 40.0, 300.0, 1024.0], n, p=[.15, .15, .3, .3, .06, .03, .01])
    s["sy"] = rng.choice([1.0, 2.0, 6.0, 25.0, 1024.0], n, p=[.2, .3, .4, .09, .01])
    s["rot"] = rng.uniform(-7, 7, n)
    s["color"] = rng.uniform(0, 1, (n, 3))
    s["opacity"] = rng.uniform(0.1, 1.0, n)
    return s


# ---------------------------------------------------------------------------------------------
# integer / index kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 63, 2048, 2049, 1_000_003])
def test_exclusive_scan(n):
    L = S2D.load_library()
    rng = np.random.default_rng(n)
    a = rng.integers(0, 50, n, dtype=np.uint32)
    want = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32) if n else a.copy()
    got = a.copy()
    import ctypes as C
    tot = C.c_uint64()
    assert L.s2d_test_exclusive_scan(0, got.ctypes.data_as(C.c_void_p), n, C.byref(tot)) == 0
    assert np.array_equal(got, want)
    assert tot.value == int(a.sum())


@pytest.mark.parametrize("n,bits", [(0, 8), (1, 8), (4095, 8), (4096, 16), (4097, 16), (300_000, 9), (2_000_000, 16), (777_777, 18)])
def test_radix_sort_stable(n, bits):
    L = S2D.load_library()
    import ctypes as C
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 1 << bits, n, dtype=np.uint32)
    if n > 10:
        keys[: n // 3] = keys[0]  # long runs of one tile
    vals = np.arange(n, dtype=np.uint32)  # emission order
    order = np.argsort(keys, kind="stable")
    k, v = keys.copy(), vals.copy()
    assert L.s2d_test_sort_pairs(0, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), n, bits) == 0
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])  # stability: equal keys keep emission (= splat index) order


def test_trig_bitwise():
    L = S2D.load_library()
    import ctypes as C
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-119.9, 119.9, 400_000), rng.uniform(0, np.pi, 400_000),
                         rng.standard_normal(10_000) * 1e-3]).astype(np.float32)
    xs = np.concatenate([xs, np.array([0x4255b0a9, 0x418a3adb, 0xc255b0a9, 0xc18a3adb], dtype=np.uint32).view(np.float32)])
    s = np.empty_like(xs)
    c = np.empty_like(xs)
    assert L.s2d_test_sincos(0, xs.ctypes.data_as(C.c_void_p), len(xs), s.ctypes.data_as(C.c_void_p),
                             c.ctypes.data_as(C.c_void_p)) == 0
    OL = O.lib()
    idx = np.concatenate([rng.choice(len(xs), 100_000, replace=False), np.arange(len(xs) - 4, len(xs))])
    ws = np.array([OL.s2do_sinf(float(xs[i])) for i in idx], dtype=np.float32)
    wc = np.array([OL.s2do_cosf(float(xs[i])) for i in idx], dtype=np.float32)
    assert np.array_equal(ws.view(np.uint32), s[idx].view(np.uint32))
    assert np.array_equal(wc.view(np.uint32), c[idx].view(np.uint32))


@pytest.mark.parametrize("W,H,n", [(268, 213, 2000), (535, 426, 50000), (4096, 4096, 100000)])
def test_init_bitwise(W, H, n):
    want = np.zeros(n, dtype=O.SPLAT_DTYPE)
    O.lib().s2do_init(want.ctypes.data, None, n, W, H)
    with S2D.Trainer(W, H, n) as t:
        t.init()
        got = t.get_splats()
        ad, b1, b2, it = t.get_adam()
    assert got.tobytes() == want.tobytes()
    assert not ad.view(np.float32).any() and b1 == 1 and b2 == 1 and it == 0


# ---------------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------------
def test_forward_it0_matches_known_answer():
    """N=1024 on the mini image: the framebuffer hash recorded from the verbatim reference."""
    with S2D.Trainer(268, 213, 1024, count_pairs=True) as t:
        t.set_target(mini_target())
        t.init()
        t.forward()
        img = t.get_image()
        st = t.stats()
    assert hashlib.sha256(img.tobytes()).hexdigest()[:16] == "6f025c573a78c6b8"
    assert abs(float(img[..., :3].astype(np.float64).sum()) - 85228.310741) < 5e-6
    assert st["fwd_active"] == 1004941 or round(st["fwd_active"] / 1e6, 3) == 1.005


@pytest.mark.parametrize("n,steps", [(1, 0), (1024, 0), (2000, 0), (2000, 7), (1024, 40)])
def test_forward_bitwise_mini(n, steps):
    o, t = make_pair(mini_target(), n, steps, count_pairs=True)
    c = O.Counters()
    want = o.forward(counters=c).copy()
    t.forward()
    got = t.get_image()
    st = t.stats()
    t.close()
    assert got.tobytes() == want.tobytes()
    assert st["fwd_active"] == c.active          # same pixels did work
    assert st["fwd_visited"] <= c.visited        # tile retirement can only skip visits of dead pixels


def test_forward_bitwise_native_50k():
    tgt = O.target_rgba32f(O.load_s2di(FULL))
    o, t = make_pair(tgt, 50000, 2)
    want = o.forward().copy()
    t.forward()
    got = t.get_image()
    t.close()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("W,H,n,seed", [(96, 80, 300, 3), (33, 17, 64, 4), (16, 16, 40, 5), (130, 50, 500, 6), (1, 1, 5, 7)])
def test_forward_bitwise_adversarial(W, H, n, seed):
    """Thin / huge / rotated / off-centre splats, low opacities, image sizes that are not tile multiples."""
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = random_splats(n, W, H, seed)
    want = o.forward().copy()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
        t.forward()
        got = t.get_image()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("seed", range(64))
def test_forward_and_gradients_random_sweep(seed):
    """Differential sweep: random image sizes (down to one pixel, not tile multiples), splat counts (down to none),
    adversarial splats, and a random variant of the path -- one set of lists (both builders), index ranges with a small pair
    budget, a row slab, fp16 images, deterministic sums.  Forward bit-exact against the oracle (on the slab's rows), gradients
    within the three bars."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 220))
    n = int(rng.choice([0, 1, 3, 40, 300, 700]))
    variant = ["plain", "generic", "chunks", "slab", "fp16", "det", "chunks+det", "slab+chunks"][seed % 8]
    kw = {}
    if "generic" in variant:
        kw["generic_binning"] = True
    if "chunks" in variant:
        kw["chunk_pairs"] = int(rng.choice([1, 50, 400]))
    if "det" in variant:
        kw["deterministic"] = True
    if "fp16" in variant:
        kw["fp16_images"] = True
    r0, r1 = 0, H
    if "slab" in variant and H > 16:
        r0 = int(rng.integers(0, (H - 1) // 16 + 1)) * 16
        r0 = min(r0, ((H - 1) // 16) * 16)
        r1 = min(H, r0 + 16 * int(rng.integers(1, 5)))
        kw.update(row_begin=r0, row_end=r1)
    tgt = O.synthetic_target(W, H)
    if "fp16" in variant:
        tgt = _fp16(tgt)
    s = random_splats(n, W, H, 2000 + seed)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = s
    want = o.forward(r0, r1).copy()
    with S2D.Trainer(W, H, n, **kw) as t:
        t.set_target(tgt)
        t.set_splats(s)
        t.forward()
        got = t.get_image()
        if "fp16" in variant:
            want = _fp16(want)
            o.image0[:] = want          # the backward pass reads the framebuffer as stored
        assert got[r0:r1].tobytes() == want[r0:r1].tobytes(), (W, H, n, variant, kw)
        t.backward()
        g = t.get_grads()
    if n:
        w32, dsum, dabs = o.backward_stats(r0, r1)
        if (dabs > 0).any():
            O.grad_bars(g.view(np.float32), w32.view(np.float32), dsum, dabs, REL)
        else:                       # no splat reaches a live pixel of these rows
            assert not g.view(np.float32).any()


@pytest.mark.parametrize("seed", range(16))
def test_training_steps_random_sweep(seed):
    """Whole iterations (s2d_step: fused forward + backward, Adam, MSE) from init() on random image sizes and splat counts,
    with "Optimize opacity" on or off and a random variant of the path: the MSE trace follows the oracle's -- iteration 0 to
    1e-9 (same framebuffer, double sum in another order), the next ones to 2e-5 -- and the parameters after 5 steps agree to
    the update bar of the single-step test, summed over the steps."""
    rng = np.random.default_rng(5000 + seed)
    W, H = int(rng.integers(40, 400)), int(rng.integers(40, 300))
    n = int(rng.choice([5, 200, 1500]))
    opacity = bool(seed & 1)
    variant = ["plain", "generic", "chunks", "det"][(seed >> 1) % 4]
    kw = {"generic_binning": True} if variant == "generic" else {"chunk_pairs": int(rng.choice([30, 2000]))} if variant == "chunks" else \
         {"deterministic": True} if variant == "det" else {}
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n, optimize_opacity=opacity)
    want = np.array([o.step()[1] for _ in range(5)])
    with S2D.Trainer(W, H, n, **kw) as t:
        t.optimize_opacity = opacity
        t.set_target(tgt)
        t.init()
        got = t.step(5)
        sp = t.get_splats().view(np.float32).reshape(-1, 9).astype(np.float64)
    assert abs(got[0] - want[0]) <= 1e-9 * want[0], (W, H, n, variant, opacity)
    np.testing.assert_allclose(got, want, rtol=2e-5)
    ws = o.splats.view(np.float32).reshape(-1, 9).astype(np.float64)
    assert np.abs(sp - ws).max() <= 5 * 1e-3 * 0.05 + 1e-5 * np.abs(ws).max(), np.abs(sp - ws).max()


@pytest.mark.parametrize("seed", range(4))
def test_garbage_parameters_neither_fault_nor_hang(seed):
    """s2d_set_splats is handed whatever the caller has: huge, tiny, zero and negative scales, positions far off the image,
    infinities and NaNs.  Outside the reference's contract (its finite guard abort()s such a state after the next Adam step,
    main.cpp:752-785) -- but the kernels must stay inside their buffers and return: forward, backward and a step complete, the
    step reports S2D_E_NONFINITE or succeeds, and the library works normally afterwards."""
    rng = np.random.default_rng(900 + seed)
    W, H, n = 300, 200, 1500
    s = random_splats(n, W, H, 50 + seed)
    bad = rng.choice(n, 300, replace=False)
    pool = np.array([0.0, -1.0, -1e30, 1e30, 1e-30, np.inf, -np.inf, np.nan, 3.4e38, 1e-45, 65536.0, -0.0], dtype=np.float32)
    raw = s.view(np.float32).reshape(n, 9)
    for i in bad:
        for k in rng.choice(9, int(rng.integers(1, 5)), replace=False):
            raw[i, k] = rng.choice(pool)
    tgt = O.synthetic_target(W, H)
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(s)
        t.forward()
        img = t.get_image()
        t.backward()
        g = t.get_grads()
        assert img.shape == (H, W, 4) and g.shape == (n,)
        try:
            t.step(2)
        except S2D.S2DError as e:
            assert e.code == 3, e           # S2D_E_NONFINITE: where the reference would abort()
    # the library is still in working order afterwards
    with S2D.Trainer(64, 64, 50) as t:
        t.set_target_synthetic()
        t.init()
        assert np.isfinite(t.step(3)).all()


def test_forward_no_splats():
    with S2D.Trainer(40, 30, 0) as t:
        t.set_target(O.synthetic_target(40, 30))
        t.forward()
        img = t.get_image()
    assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)


def test_tile_lists_sorted_and_complete():
    o, t = make_pair(mini_target(), 2000, 3)
    t.forward()
    tx, ty, off, lst = t.tile_lists()
    t.close()
    assert off[0] == 0 and off[-1] == len(lst) and np.all(np.diff(off.astype(np.int64)) >= 0)
    # ascending splat index inside every tile == the reference's blend order (main.cpp:419)
    member = set()
    for tile in range(tx * ty):
        seg = lst[off[tile]:off[tile + 1]].astype(np.int64)
        assert np.all(np.diff(seg) > 0)
        member.update((tile, int(i)) for i in seg)
    # completeness: every pixel the oracle's loops visit belongs to a tile that lists the splat
    L = O.lib()
    import ctypes as C
    one = np.zeros(1, dtype=O.SPLAT_DTYPE)
    img = np.zeros((o.H, o.W, 4), dtype=np.float32)
    for i in range(0, o.n, 37):
        one[0] = o.splats[i]
        L.s2do_forward_rows(one.ctypes.data, 1, o.W, o.H, 0, o.H, img.ctypes.data, None)
        ys, xs = np.nonzero(img[..., 0] + img[..., 1] + img[..., 2])
        for tile in set(((ys // 16) * tx + xs // 16).tolist()):
            assert (tile, i) in member


def _lists(W, H, n, splats, generic, **kw):
    with S2D.Trainer(W, H, n, generic_binning=generic, **kw) as t:
        t.set_target_synthetic()
        t.set_splats(splats)
        t.forward()
        tx, ty, off, lst = t.tile_lists()
        return tx, ty, off.copy(), lst.copy(), t.get_image()


@pytest.mark.parametrize("case", ["init_2048", "wide_5000", "adversarial", "slab", "one_tile_row", "one_tile", "empty"])
def test_two_level_tile_lists_equal_the_sorted_pairs(case):
    """The two builders of the per-tile lists -- (splat, tile row) entries sorted by row, then a counting sort by column per
    row (s2d_tilelists.hip), and all (tile, splat) pairs radix-sorted by tile (S2D_CFG_GENERIC_BINNING) -- give the same
    offsets and the same lists, word for word: init() scenes, more than 256 tile columns, splats from one pixel to the whole
    image (a chunk whose pairs do not fit the staging buffer), row slabs, a single tile row, a single tile, no splat on
    the image.  (Every list ascending in splat index: test_tile_lists_sorted_and_complete.)"""
    kw = {}
    if case == "init_2048":
        W, H, n = 2048, 2048, 250_000
        with S2D.Trainer(W, H, n) as t:
            t.init()
            sp = t.get_splats()
    elif case == "wide_5000":
        W, H, n = 5000, 700, 60_000            # 313 tile columns: the 512-column / 512-entry instantiation
        sp = random_splats(n, W, H, 11)
    elif case == "adversarial":
        W, H, n = 1500, 1100, 6000
        sp = random_splats(n, W, H, 5)
        rng = np.random.default_rng(5)
        sp["sx"] = rng.choice([1.0, 2.0, 8.0, 60.0, 400.0, 1024.0], n)   # many image-covering rectangles in every chunk
        sp["sy"] = rng.choice([1.0, 3.0, 9.0, 80.0, 1024.0], n)
        sp["pos"][:200] = [-500.0, -500.0]                                   # no footprint at all
    elif case == "slab":
        W, H, n = 1024, 1024, 40_000
        sp = random_splats(n, W, H, 3)
        kw = dict(row_begin=320, row_end=592)
    elif case == "one_tile_row":
        W, H, n = 900, 16, 3000
        sp = random_splats(n, W, H, 4)
    elif case == "one_tile":
        W, H, n = 13, 9, 500
        sp = random_splats(n, W, H, 6)
    else:
        W, H, n = 300, 200, 64
        sp = random_splats(n, W, H, 7)
        sp["pos"][:] = [-4000.0, -4000.0]
    a = _lists(W, H, n, sp, False, **kw)
    b = _lists(W, H, n, sp, True, **kw)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]), "tile offsets differ"
    assert np.array_equal(a[3], b[3]), "tile lists differ"
    assert a[4].tobytes() == b[4].tobytes()
    if case != "empty":
        assert len(a[3]) > 0


# ---------------------------------------------------------------------------------------------
# backward / Adam / MSE
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,steps,opacity", [(1024, 0, False), (2000, 0, False), (2000, 5, False), (1024, 30, True)])
def test_backward_parity_mini(n, steps, opacity):
    o, t = make_pair(mini_target(), n, steps, opacity, count_pairs=True)
    o.forward()
    c = O.Counters()
    o.backward(counters=c)
    want_mse = o.mse()
    t.forward()
    t.backward()
    got = t.get_grads()
    got_mse = t.mse()
    st = t.stats()
    t.close()
    grad_check(got, o)
    assert st["bwd_active"] == c.active
    assert abs(got_mse - want_mse) <= 1e-9 * want_mse


def test_backward_parity_adversarial():
    W, H, n = 96, 80, 300
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = random_splats(n, W, H, 3)
    o.forward()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
        t.forward()
        t.backward()
        got = t.get_grads()
    grad_check(got, o)


@pytest.mark.parametrize("opacity", [False, True])
def test_single_step_from_identical_state(opacity):
    """fwd + bwd + Adam + clamps from the same state (the parity gate of BASELINE.md §3)."""
    o, t = make_pair(mini_target(), 2000, 4, opacity)
    before = o.splats.view(np.float32).reshape(-1, 9).copy()
    st, want_mse = o.step()
    got_mse = t.step(1)[0]
    got = t.get_splats().view(np.float32).reshape(-1, 9).astype(np.float64)
    ad, b1, b2, it = t.get_adam()
    t.close()
    want = o.splats.view(np.float32).reshape(-1, 9).astype(np.float64)
    assert st == 0
    assert abs(got_mse - want_mse) <= 1e-9 * want_mse
    # Adam normalises every update to ~lr whatever the gradient's size, so the bar is on the UPDATE: the two
    # updates must agree to STEP_REL = 1e-4 of lr (beyond one ulp of the parameter, see step_delta_error).  A gradient that
    # differs by eps relative moves the update by up to ~10 eps * lr (m_hat / sqrt(v_hat) <= ~10 |g| / sqrt(v)).
    err = O.step_delta_error(before, got, want)
    assert err.max() <= STEP_REL, err.max()
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max() <= REL
    assert b1 == o.beta1t[0] and b2 == o.beta2t[0] and it == o.iterations
    wa = o.adams.view(np.float32).reshape(-1, 18).astype(np.float64)
    ga = ad.view(np.float32).reshape(-1, 18).astype(np.float64)
    scale = np.sqrt((wa ** 2).mean(axis=0)) + 1e-30
    assert (np.abs(ga - wa) / np.maximum(np.abs(wa), 1e-3 * scale[None, :])).max() <= 1e-3
    if not opacity:
        assert np.all(got[:, 8] == want[:, 8])  # opacity untouched when the checkbox is off (main.cpp:735)


def test_mse_trace_short_run_matches_reference_print():
    """First iterations of the as-shipped configuration print the same line as the reference (main.cpp:807)."""
    with S2D.Trainer(268, 213, 1024) as t:
        t.set_target(mini_target())
        t.init()
        tr = t.step(12)
    want = [5934.9042, 4659.3289, 3634.5384, 2840.9659, 2253.0626, 1839.7870, 1567.4046, 1401.9065,
            1311.3069, 1267.7320, 1248.9938, 1244.4892]
    # iteration 0 is pinned by a bit-exact forward; later ones inherit fp32 summation-order noise of the
    # gradients through Adam (the trajectory is chaotic: SURVEY.md §7 hard part 2)
    assert "%.4f" % tr[0] == "%.4f" % want[0]
    np.testing.assert_allclose(tr, want, rtol=2e-5)


def test_training_converges_like_reference():
    """300 iterations: PSNR band of the reference's run (84.76 MSE @ it 299 -> 28.85 dB)."""
    with S2D.Trainer(268, 213, 1024) as t:
        t.set_target(mini_target())
        t.init()
        tr = t.step(300)
    psnr = 10 * np.log10(255.0 ** 2 / tr[299])
    assert abs(psnr - 28.85) < 0.25, psnr
    assert abs(tr[100] - 219.3069) / 219.3069 < 0.02


def test_nonfinite_guard_reports_status():
    n = 8
    tgt = O.synthetic_target(32, 32)
    with S2D.Trainer(32, 32, n) as t:
        t.set_target(tgt)
        t.init()
        s = t.get_splats()
        s["rot"][3] = np.nan
        t.set_splats(s)
        with pytest.raises(S2D.S2DError) as ei:
            t.step(1)
        assert ei.value.code == 3  # S2D_E_NONFINITE
        assert t.stats()["first_nonfinite_iteration"] == 0


def test_nonfinite_stops_the_queue_where_the_reference_aborts():
    """The reference abort()s right after the Adam step that produced a non-finite parameter (main.cpp:752-785).
    s2d_step queues many iterations without a host round trip, so the kernels of the later ones must do nothing:
    parameters stay as that Adam step left them, the MSE trace ends with that iteration's value (NaN afterwards),
    and new parameters (set_splats) clear the condition."""
    import ctypes as C
    tgt = mini_target()
    n = 500
    with S2D.Trainer(268, 213, n) as t:
        t.set_target(tgt)
        t.init()
        t.step(3)
        ad, b1, b2, it = t.get_adam()
        assert it == 3
        ad["mv"][7, 4, 0] = np.inf  # first moment of splat 7's rot: its next update is non-finite
        t.set_adam(ad, b1, b2, it)
        mse = np.zeros(6)
        rc = t.L.s2d_step(t._h, 6, 0, mse.ctypes.data_as(C.c_void_p))
        assert rc == 3
        assert t.stats()["first_nonfinite_iteration"] == 3
        assert np.isfinite(mse[0]) and np.isnan(mse[1:]).all()   # iteration 3 printed, 4.. never ran
        s_fail = t.get_splats()
        assert not np.isfinite(s_fail["rot"][7])
        with pytest.raises(S2D.S2DError):
            t.step(2)                                             # still stopped ...
        assert t.get_splats().tobytes() == s_fail.tobytes()       # ... and nothing moved
        # a reference run from the same state reaches exactly these parameters at its abort()
        o, t2 = make_pair(tgt, n, 3)
        t2.close()
        o.adams["mv"][7, 4, 0] = np.inf
        st, m = o.step()
        assert st == 1 and abs(m - mse[0]) <= 2e-5 * m
        ok = np.isfinite(s_fail.view(np.float32)) & np.isfinite(o.splats.view(np.float32))
        assert ok.sum() == n * 9 - 1
        d = np.abs(s_fail.view(np.float32)[ok] - o.splats.view(np.float32)[ok])
        assert d.max() <= 5e-5                                    # one Adam step of 0.05 from identical state
        ad2, b1f, b2f, itf = t.get_adam()
        assert itf == 4 and b1f == o.beta1t[0] and b2f == o.beta2t[0]   # counters stand where the run stopped
        s_fail["rot"][7] = 0.0
        ad2["mv"][7, 4, :] = 0.0
        t.set_splats(s_fail)
        t.set_adam(ad2, b1f, b2f, itf)
        assert np.isfinite(t.step(2)).all()


def test_call_order_errors():
    with S2D.Trainer(32, 32, 4) as t:
        with pytest.raises(S2D.S2DError) as ei:
            t.forward()  # no target
        assert ei.value.code == 5
        t.set_target(O.synthetic_target(32, 32))
        t.init()
        with pytest.raises(S2D.S2DError):
            t.backward()  # no forward yet
    with pytest.raises(S2D.S2DError):
        S2D.Trainer(32, 32, 4, row_begin=8, row_end=32)  # slab must start on a tile row


# ---------------------------------------------------------------------------------------------
# cached tile lists
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lr,interval", [(0.0, 8), (4.0, 8), (0.0, 0), (4.0, 0)])
def test_cached_tile_lists_stay_exact(lr, interval):
    """Tile lists are re-used for several iterations (rebin_interval > 1).  Lists are supersets and inclusion
    is decided per pixel from the CURRENT parameters, so every iteration's framebuffer must still be bit-exact
    against the oracle run on the GPU's current splats.  lr = 4 moves splats ~4 px per step, far outside the
    binning margin: the on-device containment check must then force a rebuild before the raster runs."""
    tgt = mini_target()
    o = O.OracleTrainer(tgt, 2000)
    with S2D.Trainer(268, 213, 2000, rebin_interval=interval, training_rate=lr) as t:
        t.set_target(tgt)
        t.init()
        for k in range(24):
            t.forward()
            img = t.get_image()
            o.splats[:] = t.get_splats().view(O.SPLAT_DTYPE)
            assert img.tobytes() == o.forward().tobytes(), k
            t.backward()
            t.adam_step()
        st = t.stats()
    if lr == 0.0:
        assert st["rebins"] < 24 // 2   # lists really were re-used
    else:
        assert st["rebins"] > 24 // 8   # unscheduled rebuilds happened


@pytest.mark.parametrize("world", [2, 3])
def test_cached_tile_lists_stay_exact_on_row_slabs(world):
    """The same on row-slab contexts, where the fused Adam kernel skips the projection of splats whose 3-sigma
    circle cleared the slab: with training_rate = 4 a splat leaves the slab in one step while the re-used lists still
    name it, and the raster must then see an EMPTY record, not the stale one (a ghost splat)."""
    D = importlib.import_module("2dgaussiansplatting_amd.distributed")
    tgt = mini_target()
    o = O.OracleTrainer(tgt, 2000)
    for rank in range(world):
        r0, r1 = D.slab_rows(213, rank, world)
        with S2D.Trainer(268, 213, 2000, row_begin=r0, row_end=r1, rebin_interval=8, training_rate=4.0) as t:
            t.set_target(tgt)
            t.init()
            for k in range(16):
                t.forward()
                img = t.get_image()
                o.splats[:] = t.get_splats().view(O.SPLAT_DTYPE)
                assert img[r0:r1].tobytes() == o.forward(r0, r1)[r0:r1].tobytes(), (rank, k)
                t.backward()
                g = t.get_grads()
                o.backward(r0, r1)
                # gradients of the slab's rows only: same bars as everywhere (no ghost contributions)
                d = np.abs(g.view(np.float32).astype(np.float64) - o.dsplats.view(np.float32))
                assert d.max() <= 1e-3 * max(1.0, np.abs(o.dsplats.view(np.float32)).max()), (rank, k)
                t.adam_step()


# ---------------------------------------------------------------------------------------------
# row slabs (the multi-GPU partition) and the C++ host loop
# ---------------------------------------------------------------------------------------------
def test_row_slab_contexts_add_up_to_the_full_image():
    """Two slab contexts on one GPU: framebuffers tile the image bit-exactly, partial gradients and squared
    errors add up to the single-context result (what the RCCL all-reduce sums across GPUs)."""
    D = importlib.import_module("2dgaussiansplatting_amd.distributed")
    tgt = mini_target()
    o, full = make_pair(tgt, 2000, 3)
    full.forward(); full.backward()
    img_full, g_full, mse_full = full.get_image(), full.get_grads().view(np.float32).reshape(-1, 9), full.mse()
    full.close()
    o.forward()
    _, dsum, dabs = o.backward_stats()
    img = np.zeros_like(img_full)
    g = np.zeros((2000, 9), dtype=np.float64)
    mse = 0.0
    for rank in range(2):
        r0, r1 = D.slab_rows(o.H, rank, 2)
        with S2D.Trainer(o.W, o.H, 2000, row_begin=r0, row_end=r1) as t:
            t.set_target(tgt)
            t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
            t.forward(); t.backward()
            part = t.get_image()
            assert not part[:r0].any() and not part[r1:].any()
            img[r0:r1] = part[r0:r1]
            g += t.get_grads().view(np.float32).reshape(-1, 9)
            mse += t.mse()
    assert img.tobytes() == img_full.tobytes()
    assert abs(mse - mse_full) <= 1e-12 * mse_full
    nz = dabs > 0
    assert (np.abs(g - dsum)[nz] / dabs[nz]).max() <= 1e-6
    assert (np.abs(g - g_full)[nz] / dabs[nz]).max() <= 1e-6


def test_cpp_host_loop_prints_the_reference_trace():
    """2dgaussiansplatting_amd/host/splat2d_train.cpp: the headless main() loop, as-shipped configuration."""
    import subprocess
    exe = S2D._build.build_host_program()
    r = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "12"], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "0 itr, mse 5934.9042"       # main.cpp:807, SURVEY Appendix C
    want = [5934.9042, 4659.3289, 3634.5384, 2840.9659, 2253.0626, 1839.7870, 1567.4046, 1401.9065,
            1311.3069, 1267.7320, 1248.9938, 1244.4892]
    got = [float(l.split("mse")[1]) for l in lines]
    assert [int(l.split()[0]) for l in lines] == list(range(12))
    np.testing.assert_allclose(got, want, rtol=2e-5)
    # opacity checkbox ticked after the first frame (the survey's second known-answer trace), batched steps
    r = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "11", "--batch", "4",
                        "--optimize-opacity", "--opacity-from", "1"], capture_output=True, text=True, check=True)
    got = [float(l.split("mse")[1]) for l in r.stdout.strip().splitlines()]
    assert abs(got[10] - 1145.5531) / 1145.5531 < 1e-4


def test_backward_skip_opacity_grad_flag():
    """S2D_BWD_SKIP_OPACITY_GRAD leaves dSplats.opacity at zero and changes nothing else (main.cpp:704 vs :735)."""
    o, t = make_pair(mini_target(), 2000, 3)
    t.forward(); t.backward(skip_opacity_grad=True)
    lean = t.get_grads().view(np.float32).reshape(-1, 9).astype(np.float64)
    t.close()
    o.forward()
    _, dsum, dabs = o.backward_stats()
    assert np.all(lean[:, 8] == 0)
    nz = dabs[:, :8] > 0
    assert (np.abs(lean[:, :8] - dsum[:, :8])[nz] / dabs[:, :8][nz]).max() <= 1e-6
    # s2d_step without the opacity flag uses the lean kernel; with it, the full one: one step each still matches the oracle
    for opacity in (False, True):
        o2, t2 = make_pair(mini_target(), 2000, 3, opacity)
        o2.step()
        t2.step(1)
        got = t2.get_splats().view(np.float32).reshape(-1, 9).astype(np.float64)
        t2.close()
        want = o2.splats.view(np.float32).reshape(-1, 9).astype(np.float64)
        assert (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max() <= REL


# ---------------------------------------------------------------------------------------------
# more edge cases
# ---------------------------------------------------------------------------------------------
def _run_pair(W, H, splats, tgt=None, check_grads=True):
    tgt = O.synthetic_target(W, H) if tgt is None else tgt
    n = len(splats)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = splats
    want = o.forward().copy()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
        t.forward()
        got = t.get_image()
        assert got.tobytes() == want.tobytes()
        if check_grads:
            t.backward()
            grad_check(t.get_grads(), o)
        return t.stats()


def test_image_covering_splats_grow_the_pair_buffers():
    """Every splat covers every tile (sigma = 300 on 640x480): the pair count (n * tiles) outgrows the initial
    capacity, lists are 1200 entries per tile, low opacity keeps all of them alive."""
    W, H, n = 640, 480, 1200
    rng = np.random.default_rng(21)
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    s["pos"][:, 0] = rng.uniform(0, W - 1, n)
    s["pos"][:, 1] = rng.uniform(0, H - 1, n)
    s["sx"] = rng.uniform(250, 400, n)
    s["sy"] = rng.uniform(250, 400, n)
    s["rot"] = rng.uniform(0, np.pi, n)
    s["color"] = rng.uniform(0, 1, (n, 3))
    s["opacity"] = 0.1
    st = _run_pair(W, H, s, check_grads=False)
    assert st["pairs_binned"] == n * 40 * 30
    assert st["pairs_capacity"] >= st["pairs_binned"]


@pytest.mark.parametrize("budget,kw", [(3000, {}), (700, {}), (3000, {"fp16_images": True}), (2500, {"generic_binning": True}),
                                       (3000, {"row_begin": 64, "row_end": 160})])
def test_index_range_rendering_equals_one_set_of_lists(budget, kw):
    """Scenes whose (tile, splat) pairs exceed the pair budget are rendered by index ranges of the splats (front to back,
    like main.cpp:419 / :552, with the per-pixel colour and T carried from range to range).  Forced here on the mini
    scene with a budget of a few thousand pairs (default 2^30): the framebuffer must be the unchunked one bit for bit --
    the cut changes no operation -- deterministic gradients too (a splat's partial sums all lie in its own range), the
    gradients meet the oracle's bars, and whole training iterations give the same trace and parameters."""
    tgt = mini_target()
    n = 2000
    o = O.OracleTrainer(tgt, n)
    for _ in range(3):
        o.step()
    res = {}
    for chunked in (False, True):
        with S2D.Trainer(268, 213, n, deterministic=True, chunk_pairs=budget if chunked else None, **kw) as t:
            t.set_target(tgt)
            t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
            t.set_adam(o.adams.view(S2D.ADAM_DTYPE), o.beta1t[0], o.beta2t[0], o.iterations)
            t.forward()
            img = t.get_image()
            t.backward()
            g = t.get_grads()
            mse = t.mse()
            t.adam_step()
            tr = t.step(4)                       # fused forward + backward launches
            t.forward_backward()
            g2 = t.get_grads()
            t.adam_step()
            res[chunked] = (img.tobytes(), g.tobytes(), mse, tr.tobytes(), g2.tobytes(), t.get_splats().tobytes(), t.stats()["rebins"])
            if chunked and not kw:
                want = o.forward()
                assert img.tobytes() == want.tobytes()
                grad_check(g, o)
    for k, name in enumerate(["image", "gradients", "mse", "trace of s2d_step", "gradients of s2d_forward_backward", "splats"]):
        assert res[False][k] == res[True][k], name
    assert res[True][6] > 4 * res[False][6]      # the ranges really were built one after the other


def test_index_range_rendering_with_atomic_gradients_and_opacity():
    """The same with the default float-atomic gradient sums and "Optimize opacity" on: bars against the oracle."""
    tgt = mini_target()
    o, t = make_pair(tgt, 1500, steps=2, opacity=True, chunk_pairs=2000)
    with t:
        t.forward()
        assert t.get_image().tobytes() == o.forward().tobytes()
        t.backward()
        grad_check(t.get_grads(), o)
        t.adam_step()
        assert o.adam() == 0
        st, want = o.step()
        got = t.step(1)[0]
        assert st == 0 and abs(got - want) <= 2e-5 * want


@pytest.mark.parametrize("W,H,n,seed", [(8208, 64, 4000, 31), (16400, 40, 3000, 32)])
def test_images_wider_than_512_tile_columns(W, H, n, seed):
    """Beyond 8192 pixels of width (512 tile columns) the two-level list builder does not apply (its row entries carry
    nine-bit column ranges, csrc/s2d_tilelists.hip) and the library takes the generic one -- all (tile, splat) pairs
    radix-sorted by tile.  Same bars as everywhere: framebuffer bit-exact against the oracle, gradients within the three
    bars, on splats that span from one tile to the whole width; and a training step follows the oracle's."""
    s = random_splats(n, W, H, seed)
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = s
    want = o.forward().copy()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(s)
        t.forward()
        assert t.get_image().tobytes() == want.tobytes()
        t.backward()
        g = t.get_grads()
        assert t.stats()["pairs_binned"] > n
    # Splats of up to 1024 px on a strip this wide sum ~1e5..1e6 terms per scalar, and the ORACLE's sequential fp32 sum is itself
    # b_ref * sum|terms| away from the exact sum of those terms (3e-5 at 8208 px, ten times its usual distance).  The bars are
    # the usual three, scaled by that: (a) the GPU within max(1e-6, b_ref / 10) of the EXACT sum (measured 7e-7..9e-7 against a
    # b_ref of 3.2e-5, run-to-run: float atomics); (b) no further from it than the reference's own order; (c) against the
    # oracle's fp32 value -- floor 0.02 * sum|terms| -- within max(1e-4, 10 * b_ref) (measured 1.1e-4).
    w32, dsum, dabs = o.backward_stats()
    gg = g.view(np.float32).reshape(-1, 9).astype(np.float64)
    ww = w32.view(np.float32).reshape(-1, 9).astype(np.float64)
    nz = dabs > 0
    assert np.all(gg[~nz] == 0)
    b_ref = float((np.abs(ww - dsum)[nz] / dabs[nz]).max())
    a = float((np.abs(gg - dsum)[nz] / dabs[nz]).max())
    c = float((np.abs(gg - ww)[nz] / np.maximum(np.abs(ww[nz]), 0.02 * dabs[nz])).max())
    assert a <= max(1e-6, 0.1 * b_ref) and a <= b_ref, (a, b_ref)
    assert c <= max(REL, 10.0 * b_ref), (c, b_ref)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = s
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(s)
        got = t.step(3)
    want = [o.step()[1] for _ in range(3)]
    # (one-pixel-thin and 1024-px splats side by side: a last-place difference in a gradient sum flips a pixel's inclusion
    # a step later, so the bar on the third value is 1e-4; measured 1e-6 .. 2.5e-5 from run to run)
    assert abs(got[0] - want[0]) <= 1e-9 * want[0]
    np.testing.assert_allclose(got, want, rtol=1e-4)


def test_low_opacity_deep_stacks():
    """opacity 0.1: ~50 splats contribute to every pixel before the 1/256 cut-off; gradients of deep stacks."""
    W, H, n = 160, 120, 3000
    rng = np.random.default_rng(22)
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    s["pos"][:, 0] = rng.uniform(0, W - 1, n)
    s["pos"][:, 1] = rng.uniform(0, H - 1, n)
    s["sx"] = rng.uniform(4, 12, n)
    s["sy"] = rng.uniform(4, 12, n)
    s["rot"] = rng.uniform(0, np.pi, n)
    s["color"] = rng.uniform(0, 1, (n, 3))
    s["opacity"] = rng.choice([0.1, 0.15, 0.3], n)
    _run_pair(W, H, s, tgt=O.target_rgba32f(O.load_s2di(MINI))[:H, :W].copy())


def test_minimum_size_splats_and_pixel_centres():
    """sx = sy = 1 (the clamp floor, main.cpp:744-745), positions exactly on pixel centres / corners / borders:
    alpha reaches exactly 1 at a centre (1 - alpha + 1e-15 = 1e-15 in the backward pass, main.cpp:628)."""
    W, H = 48, 32
    pts = [(x + dx, y + dy) for x in (0, 15, 16, 31, 47) for y in (0, 15, 16, 31) for dx, dy in ((0.5, 0.5), (0.0, 0.0))]
    pts = [(min(px, W - 1), min(py, H - 1)) for px, py in pts]
    n = len(pts)
    s = np.zeros(n, dtype=O.SPLAT_DTYPE)
    s["pos"] = np.array(pts, dtype=np.float32)
    s["sx"] = 1.0
    s["sy"] = 1.0
    s["rot"] = np.linspace(0, 3, n)
    s["color"] = np.linspace(0.1, 0.9, n)[:, None]
    s["opacity"] = 1.0
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n)
    o.splats[:] = s
    want = o.forward().copy()
    wg = o.backward().copy()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.set_splats(s.view(S2D.SPLAT_DTYPE))
        t.forward()
        assert t.get_image().tobytes() == want.tobytes()
        t.backward()
        g = t.get_grads().view(np.float32).reshape(-1, 9)
    w = wg.view(np.float32).reshape(-1, 9)
    # where alpha == 1 exactly the reference divides by 1e-15: huge but finite terms; same magnitudes on both sides
    assert np.isfinite(g).all() == np.isfinite(w).all()
    fin = np.isfinite(w) & np.isfinite(g)
    np.testing.assert_allclose(g[fin], w[fin], rtol=1e-3, atol=1e-3 * np.abs(w[fin]).max())


def test_splats_entirely_off_image_and_single_row_image():
    W, H = 200, 1
    s = random_splats(50, W, 4, 31)
    s["pos"][:, 1] = 0
    _run_pair(W, H, s)
    W, H = 64, 64
    s = random_splats(40, W, H, 32)
    s["pos"][:20] = [-500.0, 30.0]      # far outside on the left: the clamp would normally prevent this
    s["pos"][20:30] = [30.0, 5000.0]
    st = _run_pair(W, H, s)
    assert st["pairs_binned"] < 40 * 16


def test_checkpoint_roundtrip_resumes_identically():
    """State that crosses the boundary (main.cpp:272-278) is enough to resume: same forward image bit for bit."""
    tgt = mini_target()
    with S2D.Trainer(268, 213, 1500) as a:
        a.set_target(tgt)
        a.init()
        a.step(7)
        sp, (ad, b1, b2, it) = a.get_splats(), a.get_adam()
        a.forward()
        img_a = a.get_image()
        a.step(3)
        ref = a.get_splats()
    with S2D.Trainer(268, 213, 1500) as b:
        b.set_target(tgt)
        b.set_splats(sp)
        b.set_adam(ad, b1, b2, it)
        b.forward()
        assert b.get_image().tobytes() == img_a.tobytes()
        b.step(3)
        got = b.get_splats()
        assert b.stats()["iterations"] == 10
    np.testing.assert_allclose(got.view(np.float32), ref.view(np.float32), rtol=1e-4, atol=1e-4)


def test_cpp_host_checkpoint_and_ppm(tmp_path):
    """splat2d_train: PPM input, --save-checkpoint / --load-checkpoint continue the same trajectory, PPM export."""
    import subprocess
    exe = S2D._build.build_host_program()
    rgb = O.load_s2di(MINI)
    ppm = tmp_path / "mini.ppm"
    with open(ppm, "wb") as f:
        f.write(b"P6\n# decoded fixture\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes())
    ck = str(tmp_path / "ck.bin")
    out = str(tmp_path / "out.ppm")
    a = subprocess.run([exe, "--image", str(ppm), "--splats", "1024", "--iters", "6", "--save-checkpoint", ck],
                       capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert a[0] == "0 itr, mse 5934.9042"
    b = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "4", "--load-checkpoint", ck, "--out-ppm", out],
                       capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert [int(l.split()[0]) for l in b] == [6, 7, 8, 9]
    full = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "10"], capture_output=True, text=True,
                          check=True).stdout.strip().splitlines()
    np.testing.assert_allclose([float(l.split("mse")[1]) for l in a + b], [float(l.split("mse")[1]) for l in full], rtol=2e-5)
    hdr = open(out, "rb").read(15)
    assert hdr.startswith(b"P6\n268 213\n255\n") and os.path.getsize(out) == 15 + 268 * 213 * 3
    # restart button: the trace starts over from the init() state
    r = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "5", "--restart-at", "3"], capture_output=True,
                       text=True, check=True).stdout.strip().splitlines()
    assert r[3] == "0 itr, mse 5934.9042" and r[0] == "0 itr, mse 5934.9042"   # init() resets iterations (main.cpp:281)
    assert [int(l.split()[0]) for l in r] == [0, 1, 2, 0, 1] and r[4] == r[1]
    # an output file that cannot be written is an error
    bad = subprocess.run([exe, "--image", MINI, "--splats", "64", "--iters", "1", "--out-image", "/nonexistent-dir/x.ppm"],
                         capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot write" in bad.stderr


def test_against_committed_oracle_golden():
    """Same check as the live-oracle tests, against the committed bundle tests/golden/oracle_cfg1_it5.npz
    (BASELINE configs[0]: mini image, N = 2000, state after 5 iterations)."""
    z = np.load(os.path.join(O.GOLDEN, "oracle_cfg1_it5.npz"))
    with S2D.Trainer(268, 213, 2000, count_pairs=True) as t:
        t.set_target(mini_target())
        t.set_splats(z["splats"].view(S2D.SPLAT_DTYPE))
        t.set_adam(z["adams"].view(S2D.ADAM_DTYPE), z["beta1t"][0], z["beta2t"][0], int(z["iterations"]))
        t.forward()
        img = t.get_image()
        assert hashlib.sha256(img.tobytes()).hexdigest() == str(z["image_sha256"])
        assert img[100:104].tobytes() == z["image_rows_100_103"].tobytes()
        t.backward()
        g = t.get_grads().view(np.float32).reshape(-1, 9).astype(np.float64)
        assert t.stats()["bwd_active"] == int(z["active_pairs"])
        assert abs(t.mse() - float(z["mse"])) <= 1e-9 * float(z["mse"])
        dsum, dabs, w = z["grads_exact_sum"], z["grads_abs_sum"].astype(np.float64), z["grads_fp32"].astype(np.float64)
        nz = dabs > 0
        assert (np.abs(g - dsum)[nz] / dabs[nz]).max() <= 1e-6
        assert (np.abs(g - w)[nz] / np.maximum(np.abs(w[nz]), 0.02 * dabs[nz])).max() <= REL
        t.adam_step()
        got = t.get_splats().view(np.float32).reshape(-1, 9).astype(np.float64)
    want = z["splats_after_step"].view(np.float32).reshape(-1, 9).astype(np.float64)
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max() <= REL


def test_cpp_host_learning_rate_and_deterministic_flags():
    """lr as a flag of the host program (trainingRate, main.cpp:715).  splat2d_train --lr 0.02 follows the
    oracle's loop run with that rate; --deterministic runs are bitwise reproducible."""
    import subprocess
    exe = S2D._build.build_host_program()
    tgt = mini_target()
    o = O.OracleTrainer(tgt, 1024)
    want = []
    for _ in range(6):
        o.forward()
        want.append(o.mse())
        o.backward()
        assert o.adam(np.float32(0.02)) == 0
    runs = [subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", "6", "--lr", "0.02", "--deterministic"], capture_output=True,
                           text=True, timeout=300) for _ in range(2)]
    assert all(r.returncode == 0 for r in runs), runs[0].stderr[-1500:]
    got = [float(ln.split("mse")[1]) for ln in runs[0].stdout.strip().splitlines()]
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=5e-5)   # (%.4f prints)
    assert abs(got[0] - 5934.9042) < 1e-3 and got[1] > 5200      # a 0.02 step lowers the MSE less than the 0.05 one (4659.3)
    assert runs[0].stdout == runs[1].stdout


def test_cpp_host_png_target_and_overlay(tmp_path):
    """A PNG target gives the same trace as the raw fixture; --overlay writes the reference's splat debug drawing
    (main.cpp:441-485) as an image."""
    import subprocess
    from PIL import Image
    exe = S2D._build.build_host_program()
    png = str(tmp_path / "mini.png")
    subprocess.run([exe, "--convert", MINI, png], check=True)
    a = subprocess.run([exe, "--image", png, "--splats", "64", "--iters", "3", "--out-image", str(tmp_path / "o.png"),
                        "--overlay", str(tmp_path / "ov.png"), "--overlay-scale", "3"], capture_output=True, text=True, check=True)
    b = subprocess.run([exe, "--image", MINI, "--splats", "64", "--iters", "3"], capture_output=True, text=True, check=True)
    # same pixels in, same trace out (up to run-to-run float-atomic ordering in the gradients)
    ta = [float(l.split("mse")[1]) for l in a.stdout.strip().splitlines()]
    tb = [float(l.split("mse")[1]) for l in b.stdout.strip().splitlines()]
    assert ta[0] == tb[0] and len(ta) == 3
    np.testing.assert_allclose(ta, tb, rtol=1e-6)
    o = np.asarray(Image.open(tmp_path / "o.png"))
    ov = np.asarray(Image.open(tmp_path / "ov.png"))
    assert o.shape == (213, 268, 3) and ov.shape == (639, 804, 3)
    up = o.repeat(3, 0).repeat(3, 1)
    changed = (ov != up).any(axis=2)
    assert 0.01 < changed.mean() < 0.5            # lines were drawn, the picture is still there
    assert (ov[changed] == 128).all(axis=1).any() and (ov[changed] == 255).all(axis=1).any()  # grey boxes, white axes


def _checkpoint_splats(path, n):
    """The splat records of a splat2d_train checkpoint (host/splat2d_train.cpp CkptHeader: 28-byte header, then n x 36 bytes)."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"S2DC" and len(raw) == 28 + n * (36 + 72), len(raw)
    return np.frombuffer(raw, dtype=O.SPLAT_DTYPE, count=n, offset=28).copy()


@pytest.mark.parametrize("iters", [0, 10])
def test_cpp_host_overlay_vertices_are_the_references_primvertex_list(tmp_path, iters):
    """splat2d_train --overlay-vertices dumps the line segments its overlay draws: for the mini scene at iteration 0 and
    after 10 steps they must be, bit for bit, the oracle's restatement of main.cpp:419-477 (eigen_vectors_of_cov :223-234,
    axes :443-451, 16-gon :454-462, 1-sigma box :464-477) evaluated on the splats the run ended with (its checkpoint):
    floats bitwise, colours as the u8 triples.  The rasterisation itself stays covered by the property test above."""
    import subprocess
    exe = S2D._build.build_host_program()
    ck, dump = str(tmp_path / "s.ckpt"), str(tmp_path / "v.bin")
    r = subprocess.run([exe, "--image", MINI, "--splats", "1024", "--iters", str(iters), "--save-checkpoint", ck,
                        "--overlay", str(tmp_path / "ov.ppm"), "--overlay-vertices", dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    splats = _checkpoint_splats(ck, 1024)
    xyz, rgb = O.read_overlay_dump(dump)
    want_xyz, want_rgb = O.overlay_vertices(splats)
    assert xyz.shape == (1024 * O.OVERLAY_VERTICES, 3)
    assert xyz.view(np.uint32).tolist() == want_xyz.view(np.uint32).tolist()
    assert rgb.tobytes() == want_rgb.tobytes()
    if iters == 0:   # the state init() leaves (main.cpp:280-305) is the oracle's
        o = O.OracleTrainer(mini_target(), 1024)
        assert splats.tobytes() == o.splats.tobytes()


# ---------------------------------------------------------------------------------------------
# S2D_CFG_FP16_IMAGES (BASELINE configs[4]: "fp16 color / fp32 grads")
# ---------------------------------------------------------------------------------------------
def _fp16(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("n,steps", [(2000, 0), (2000, 6)])
def test_fp16_images_match_the_oracle_with_rounded_images(n, steps):
    """With fp16 image storage the GPU must equal the reference loop run on a target rounded to fp16 and with
    image0 rounded to fp16 between the forward and the backward pass (round to nearest even both sides)."""
    tgt16 = _fp16(mini_target())
    o = O.OracleTrainer(tgt16, n)
    for _ in range(steps):
        o.forward()
        o.image0[:] = _fp16(o.image0)
        o.backward()
        assert o.adam() == 0
    with S2D.Trainer(o.W, o.H, n, fp16_images=True, count_pairs=True) as t:
        t.set_target(mini_target())   # converted on the device
        t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
        t.set_adam(o.adams.view(S2D.ADAM_DTYPE), o.beta1t[0], o.beta2t[0], steps)
        t.forward()
        img = t.get_image()
        want = _fp16(o.forward())
        assert img.tobytes() == want.tobytes()          # the rounded framebuffer, bit for bit
        o.image0[:] = want
        t.backward()
        grad_check(t.get_grads(), o)
        assert abs(t.mse() - o.mse()) <= 1e-9 * o.mse()


def test_fp16_images_train_like_fp32():
    tgt = mini_target()
    res = []
    for half in (False, True):
        with S2D.Trainer(268, 213, 1024, fp16_images=half) as t:
            t.set_target(tgt)
            t.init()
            res.append(t.step(60))
    np.testing.assert_allclose(res[1], res[0], rtol=5e-3)   # fp16 colour costs a few 1e-4 of MSE, no more
    with S2D.Trainer(2048, 2048, 250_000, fp16_images=True) as t:
        t.set_target_synthetic()
        t.init()
        tr = t.step(5)
        assert np.isfinite(tr).all() and tr[-1] < tr[0]


# ---------------------------------------------------------------------------------------------
# S2D_CFG_DETERMINISTIC
# ---------------------------------------------------------------------------------------------
def test_deterministic_mode_is_bitwise_reproducible_and_still_in_parity():
    tgt = mini_target()
    finals, traces = [], []
    for interval in (0, 0, 1):   # twice with re-used lists, once rebuilding every iteration: all three identical
        with S2D.Trainer(268, 213, 2000, deterministic=True, rebin_interval=interval) as t:
            t.set_target(tgt)
            t.init()
            traces.append(t.step(25))
            finals.append(t.get_splats().tobytes())
    assert finals[0] == finals[1] == finals[2]
    assert np.array_equal(traces[0], traces[1]) and np.array_equal(traces[0], traces[2])
    # parity of the deterministic gradients against the oracle, same bars as the atomic path
    o, t = make_pair(tgt, 2000, 5, deterministic=True)
    o.forward()
    t.forward(); t.backward()
    g1 = t.get_grads()
    grad_check(g1, o)
    t.adam_step()           # re-zeroes the gradient buffer
    t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
    t.forward(); t.backward()
    assert t.get_grads().tobytes() == g1.tobytes()   # same state, same bits
    t.close()


def test_deterministic_mode_full_size_matches_atomic_mode():
    res = []
    for det in (False, True, True):
        with S2D.Trainer(2048, 2048, 250_000, deterministic=det) as t:
            t.set_target_synthetic()
            t.init()
            t.forward(); t.backward()
            res.append(t.get_grads().view(np.float32).reshape(-1, 9).astype(np.float64))
    assert np.array_equal(res[1], res[2])
    scale = np.abs(res[1]).mean(axis=0)
    assert (np.abs(res[0] - res[1]) / (np.abs(res[1]) + scale[None, :])).max() < 1e-4


def test_repeated_forward_does_not_rebuild_lists_again():
    tgt = mini_target()
    with S2D.Trainer(268, 213, 2000, training_rate=4.0) as t:   # lr 4: every step pushes splats out of their rectangles
        t.set_target(tgt)
        t.init()
        for _ in range(3):
            t.forward(); t.backward(); t.adam_step()
        t.forward()
        r1 = t.stats()["rebins"]
        img1 = t.get_image()
        t.forward()
        t.forward()
        assert t.stats()["rebins"] == r1
        assert t.get_image().tobytes() == img1.tobytes()


# ---------------------------------------------------------------------------------------------
# slab ownership (distributed.HaloStep) through the C ABI: several slab contexts on the one GPU, the ranks as threads
# ---------------------------------------------------------------------------------------------
def _run_ranks(world, W, H, n, steps, make_step, tgt=None):
    """Runs `world` slab contexts to completion; returns per-rank dicts."""
    import torch
    from thread_dist import ThreadDist
    D = importlib.import_module("2dgaussiansplatting_amd.distributed")
    res = [None] * world
    torch.cuda.init()   # torch's lazy device initialisation, here rather than by several rank threads at once

    def body(rank, dist):
        r0, r1 = D.slab_rows(H, rank, world)
        stream = torch.cuda.Stream()      # one stream per rank, shared by torch and the library (as in bench.py)
        torch.cuda.set_stream(stream)     # thread-local
        grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
        with S2D.Trainer(W, H, n, row_begin=r0, row_end=r1, deterministic=True, stream=stream.cuda_stream) as t:
            t.bind_grads(grads.data_ptr())
            if tgt is None:
                t.set_target_synthetic()
            else:
                t.set_target(tgt)
            t.init()
            step = make_step(D, t, grads, dist, rank, world, H)
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            out = {"sq": t.sqerr_trace(0, steps), "raw": t.get_splats().view(np.float32).reshape(n, 9).copy()}
            if hasattr(step, "gather_full"):
                out["full"] = step.gather_full(D.ROWS_SPLATS).cpu().numpy()
                out["adam"] = step.gather_full(D.ROWS_ADAM).cpu().numpy()
                out["mask"] = step.mask.cpu().numpy()
                out["moved"] = step.handed_over
            res[rank] = out

    ThreadDist(world).run(body)
    return res


@pytest.mark.parametrize("world,interval,margin", [(2, 1, 1.0), (3, 2, 2.0)])
def test_slab_ownership_equals_replicated_state(world, interval, margin):
    """HaloStep (ownership + halo exchange) against SlabStep (replicated state, dense all-reduce) on the same slabs,
    deterministic gradients: identical holders, complete cover, hand-overs happen, and the assembled parameters
    equal the replicated run's -- bit for bit at two ranks (a + b is the only sum either scheme forms)."""
    W, H, n, steps = 268, 213, 1500, 10
    tgt = mini_target()
    halo = _run_ranks(world, W, H, n, steps, lambda D, t, g, dist, r, w, hh: D.HaloStep(
        t, D.HipHaloOps(t, n, "cuda"), dist, r, w, hh, rehalo_interval=interval, margin_rows=margin), tgt)
    dense = _run_ranks(world, W, H, n, steps, lambda D, t, g, dist, r, w, hh: D.SlabStep(t, g, dist), tgt)
    union = np.zeros(n, dtype=np.int64)
    for q in range(world):
        union |= np.where((halo[q]["mask"] >> q) & 1, 1 << q, 0)
    assert (union != 0).all()
    first = np.array([int(u & -u).bit_length() - 1 for u in union])
    canon = np.stack([halo[q]["raw"] for q in range(world)])[first, np.arange(n)]
    for q in range(world):
        held = ((halo[q]["mask"] >> q) & 1).astype(bool)
        assert 0 < held.sum() < n                                   # ownership really is partial
        assert (halo[q]["mask"][held] == union[held]).all() and not halo[q]["mask"][~held].any()
        assert halo[q]["raw"][held].tobytes() == canon[held].tobytes()
        assert halo[q]["full"].tobytes() == canon.tobytes()         # gather_full returns the lowest holder's rows
    assert sum(h["moved"] for h in halo) > 0
    for q in range(1, world):
        assert dense[q]["raw"].tobytes() == dense[0]["raw"].tobytes()
    sq_h = sum(h["sq"] for h in halo)
    sq_d = sum(d["sq"] for d in dense)
    if world == 2:
        assert canon.tobytes() == dense[0]["raw"].tobytes()
        assert sq_h.tobytes() == sq_d.tobytes()
    else:  # a rank that does not hold a splat contributes an exact 0 to the dense sum: still the same additions
        assert canon.tobytes() == dense[0]["raw"].tobytes()
        np.testing.assert_allclose(sq_h, sq_d, rtol=1e-12)


def test_slab_ownership_default_margin_at_size():
    """Default refresh interval and margin on a workload with thousands of boundary splats: bit-identical to the
    replicated scheme on the same four slabs (both add the slabs' partials in rank order; a rank that does not hold
    a splat contributes an exact 0 there), and the MSE trace of the single-context run."""
    W, H, n, steps = 1024, 768, 60000, 72   # one refresh (every 64 iterations) besides the one at construction
    halo = _run_ranks(4, W, H, n, steps, lambda D, t, g, dist, r, w, hh: D.HaloStep(
        t, D.HipHaloOps(t, n, "cuda"), dist, r, w, hh))
    dense = _run_ranks(4, W, H, n, steps, lambda D, t, g, dist, r, w, hh: D.SlabStep(t, g, dist))
    with S2D.Trainer(W, H, n, deterministic=True) as s:
        s.set_target_synthetic(); s.init()
        ref = np.array(s.step(steps)) * (H * W * 3)
    sq = sum(h["sq"] for h in halo)
    np.testing.assert_allclose(sq, ref, rtol=1e-5)
    assert sq.tobytes() == sum(d["sq"] for d in dense).tobytes()
    assert halo[0]["full"].tobytes() == dense[0]["raw"].tobytes()
    assert sum(h["moved"] for h in halo) > 0
    held = [((h["mask"] >> q) & 1).astype(bool).mean() for q, h in enumerate(halo)]
    assert max(held) < 0.45 and sum(held) < 1.6   # a quarter of the image each, plus halos


def test_row_level_abi_calls_against_numpy():
    """s2d_rows_gather / s2d_rows_scatter / s2d_grads_combine / s2d_halo_masks / s2d_halo_commit, one by one."""
    import torch
    D = importlib.import_module("2dgaussiansplatting_amd.distributed")
    W, H, n = 268, 213, 1000
    rng = np.random.default_rng(5)
    with S2D.Trainer(W, H, n) as t:
        t.set_target(mini_target())
        t.init()
        ops = D.HipHaloOps(t, n, "cuda")
        sp = t.get_splats().view(np.float32).reshape(n, 9)
        ids = torch.from_numpy(rng.permutation(n)[:300].astype(np.int32)).cuda()
        got = ops.rows_gather(D.ROWS_SPLATS, ids).cpu().numpy()
        assert got.tobytes() == sp[ids.cpu().numpy()].tobytes()
        # scatter new Adam rows, read the whole state back through the ordinary accessor
        new = torch.from_numpy(rng.standard_normal((300, 18)).astype(np.float32)).cuda()
        ops.rows_scatter(D.ROWS_ADAM, ids, new)
        ad = t.get_adam()[0].view(np.float32).reshape(n, 18)
        assert ad[ids.cpu().numpy()].tobytes() == new.cpu().numpy().tobytes()
        untouched = np.ones(n, dtype=bool); untouched[ids.cpu().numpy()] = False
        assert not ad[untouched].any()
        # masks: the formula of s2d_halo.hip in numpy
        bounds = [0, 64, 128, 213]
        m = ops.halo_masks(bounds, 4.0).cpu().numpy()
        reach = np.float32(3.0) * np.maximum(sp[:, 2], sp[:, 3]) + np.float32(2.0) + np.float32(4.0)
        want = np.zeros(n, dtype=np.int32)
        for q in range(3):
            want |= ((sp[:, 1] + reach >= np.float32(bounds[q])) & (sp[:, 1] - reach <= np.float32(bounds[q + 1]))).astype(np.int32) << q
        assert (m == want).all() and (m != 0).all()
        # combine: grads rows = partials of ranks 0, 1 (this one), 2 added in that order
        t.forward(); t.backward()
        g = t.get_grads().view(np.float32).reshape(n, 9).copy()
        rows = torch.from_numpy(np.sort(rng.permutation(n)[:200]).astype(np.int32)).cuda()
        recv = torch.from_numpy(rng.standard_normal((400, 9)).astype(np.float32)).cuda()
        src = np.full((200, 3), -1, dtype=np.int32)
        src[:, 1] = -2
        src[:150, 0] = np.arange(150)                # rank 0 holds the first 150 rows
        src[100:, 2] = 200 + np.arange(100)          # rank 2 holds the last 100
        ops.grads_combine(rows, torch.from_numpy(src).cuda(), recv)
        got = t.get_grads().view(np.float32).reshape(n, 9)
        r, rc = rows.cpu().numpy(), recv.cpu().numpy()
        want = g.copy()
        for u, i in enumerate(r):
            acc = None
            for q in range(3):
                s = src[u, q]
                if s == -1:
                    continue
                v = g[i] if s == -2 else rc[s]
                acc = v.copy() if acc is None else (acc + v).astype(np.float32)
            want[i] = acc
        assert got.tobytes() == want.tobytes()
        # commit: only held splats are updated by Adam from here on
        mask = torch.from_numpy((np.arange(n) % 2 == 0).astype(np.int32) * 2).cuda()  # rank 1 holds the even ids
        ops.halo_commit(mask.contiguous(), 1)
        before = t.get_splats().view(np.float32).reshape(n, 9).copy()
        t.forward(); t.backward(); t.adam_step()
        after = t.get_splats().view(np.float32).reshape(n, 9)
        assert after[1::2].tobytes() == before[1::2].tobytes()
        assert (after[0::2] != before[0::2]).any()


def test_adam_fp32_quotient_mode_matches_the_oracle_in_that_mode():
    """S2D_CFG_ADAM_FP32 (the reference's MSVC evaluation of main.cpp:155): one step from identical state against the
    oracle with the same switch, to the same bar as the default double-precision form."""
    tgt = mini_target()
    try:
        O.lib().s2do_set_adam_fp32(1)
        o, t0 = make_pair(tgt, 2000, 4)
        t0.close()
        before = o.splats.view(np.float32).reshape(-1, 9).copy()
        with S2D.Trainer(o.W, o.H, 2000, adam_fp32=True) as t:
            t.set_target(tgt)
            t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
            t.set_adam(o.adams.view(S2D.ADAM_DTYPE), o.beta1t[0], o.beta2t[0], o.iterations)
            o.step()
            t.step(1)
            got = t.get_splats().view(np.float32).reshape(-1, 9)
    finally:
        O.lib().s2do_set_adam_fp32(0)
    err = O.step_delta_error(before, got, o.splats.view(np.float32).reshape(-1, 9))
    assert err.max() <= STEP_REL, err.max()


@pytest.mark.parametrize("kw", [{}, {"deterministic": True}, {"fp16_images": True}, {"deterministic": True, "fp16_images": True}])
def test_fused_forward_backward_equals_the_two_passes(kw):
    """s2d_forward_backward (one launch per tile, what s2d_step queues) against s2d_forward + s2d_backward: identical
    framebuffer, identical squared error, and gradients that are bitwise equal with deterministic sums (same terms,
    same order) and equal to summation noise with float atomics."""
    tgt = mini_target()
    res = []
    for fused in (False, True):
        o, t = make_pair(tgt, 2000, 3, **kw)
        if fused:
            t.forward_backward()
        else:
            t.forward()
            t.backward()
        res.append((t.get_image(), t.get_grads().view(np.float32).reshape(-1, 9).copy(), t.mse()))
        t.close()
    (img_a, g_a, m_a), (img_b, g_b, m_b) = res
    assert img_a.tobytes() == img_b.tobytes()
    assert m_a == m_b
    if kw.get("deterministic"):
        assert g_a.tobytes() == g_b.tobytes()
    else:
        o.image0[:] = 0
        if kw.get("fp16_images"):
            pass  # bars against the oracle are checked by test_fp16_images_*; here only fused vs separate
        scale = np.abs(g_a).max(axis=0) + 1e-30
        assert (np.abs(g_a - g_b) / scale).max() <= 1e-5


def test_step_leaves_the_image_of_its_last_iteration():
    """s2d_step stores image0 only in the last iteration of the call (nothing can observe the others): what
    s2d_get_image returns afterwards is the framebuffer rendered from the parameters BEFORE the last update, as
    uploaded at main.cpp:794."""
    tgt = mini_target()
    with S2D.Trainer(268, 213, 1500, deterministic=True) as a, S2D.Trainer(268, 213, 1500, deterministic=True) as b:
        for t in (a, b):
            t.set_target(tgt)
            t.init()
        a.step(4)
        b.step(3)
        b.forward()
        assert a.get_image().tobytes() == b.get_image().tobytes()
        b.backward()
        b.adam_step()
        assert a.get_splats().tobytes() == b.get_splats().tobytes()   # and the separate passes train identically


def test_cpp_host_multi_gpu_path_through_rccl_on_one_gpu():
    """host/splat2d_train.cpp --gpus N --exchange dense drives a multi-device handle (s2d_multi_*) in its replicated-state
    scheme: one worker thread + one context per GPU, ncclAllReduce of the N x 9 gradients between s2d_forward_backward
    and s2d_adam_step.  This box has one GPU (RCCL takes one rank per GPU), so the test sends --gpus 1 through that
    same code (S2D_TRAIN_FORCE_MULTI): RCCL
    loaded with dlopen, communicator set-up, the in-place all-reduce on the context's stream, the sum of the slabs'
    squared errors and the reference's trace line."""
    import subprocess
    exe = os.path.join(os.path.dirname(S2D.__file__), "lib", "splat2d_train")
    args = [exe, "--image", MINI, "--splats", "1024", "--iters", "12", "--batch", "4"]
    want = subprocess.run(args, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    env = dict(os.environ, S2D_TRAIN_FORCE_MULTI="1")
    p = subprocess.run(args + ["--gpus", "1", "--exchange", "dense"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    got = [ln for ln in p.stdout.strip().splitlines() if " itr, mse " in ln]
    assert got[0] == "0 itr, mse 5934.9042" and len(got) == 12
    np.testing.assert_allclose([float(l.split("mse")[1]) for l in got], [float(l.split("mse")[1]) for l in want], rtol=2e-5)
    assert "RCCL all-reduce" in p.stderr
    # more ranks than this box has GPUs: a clean error, not a crash (either scheme)
    for extra in ([], ["--exchange", "dense"]):
        p = subprocess.run(args + ["--gpus", "2"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and p.stdout.strip() == ""


def test_forward_backward_skip_image_flag():
    """S2D_FB_SKIP_IMAGE: same gradients and squared error, image0 left as it was."""
    tgt = mini_target()
    o, t = make_pair(tgt, 2000, 2, deterministic=True)
    t.forward()
    old = t.get_image()
    t.backward()
    g0, m0 = t.get_grads().tobytes(), t.mse()
    t.adam_step()
    o2, t2 = make_pair(tgt, 2000, 2, deterministic=True)
    t2.forward_backward(skip_image=True)
    assert t2.get_grads().tobytes() == g0 and t2.mse() == m0
    assert not t2.get_image().any()               # never stored in this context
    t.forward_backward(skip_image=True)           # a step later: the old frame stays
    assert t.get_image().tobytes() == old.tobytes()
    t.close(); t2.close()


@pytest.mark.parametrize("world,exchange", [(2, "halo"), (3, "halo"), (2, "dense"), (3, "dense")])
def test_cpp_host_n_ranks_sharing_the_gpu(world, exchange):
    """splat2d_train --gpus N --share-gpu: the N-rank logic of the multi-device handle under the C++ loop (one thread +
    one slab context per rank, the exchange between s2d_forward_backward and s2d_adam_step -- gradient rows of shared
    splats between their holders, or the sum of all gradients --, the sum of the slabs' squared errors) on a box with
    one GPU.  The trace must follow the single-context one (only the fp32 order of the gradient sums differs)."""
    import subprocess
    exe = os.path.join(os.path.dirname(S2D.__file__), "lib", "splat2d_train")
    args = [exe, "--image", MINI, "--splats", "1024", "--iters", "12", "--batch", "3"]
    want = subprocess.run(args, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    p = subprocess.run(args + ["--gpus", str(world), "--share-gpu", "--exchange", exchange], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    got = [ln for ln in p.stdout.strip().splitlines() if " itr, mse " in ln]
    assert len(got) == 12 and got[0] == "0 itr, mse 5934.9042"
    np.testing.assert_allclose([float(l.split("mse")[1]) for l in got], [float(l.split("mse")[1]) for l in want], rtol=2e-5)
    assert "%d ranks" % world in p.stderr
    assert ("slab ownership" if exchange == "halo" else "replicated state") in p.stderr
    for r in range(world):     # which GPU runs which rows (s2d_multi_device_info)
        assert any(ln.startswith("rank %d: device 0 (" % r) and "rows " in ln for ln in p.stderr.splitlines()), p.stderr


@pytest.mark.parametrize("world,replicated", [(2, False), (3, False), (2, True), (3, True)])
def test_multi_device_handle_ranks_sharing_the_gpu(world, replicated):
    """s2d_multi_* (several GPUs behind one handle: the fan-out SURVEY.md section 8b asks for) with all ranks on this
    box's one GPU (S2D_MULTI_SHARE_GPU), in both schemes: the first frame is bit-identical to the single context's
    (same init(), same rows), the trace follows it (only the fp32 order of the gradient sums differs), the state comes
    back through the handle, and a deterministic handle reproduces itself bit for bit."""
    tgt = mini_target()
    with S2D.Trainer(268, 213, 1500) as t:
        t.set_target(tgt)
        t.init()
        want0 = t.step(1)
        want_img = t.get_image()
        want = np.concatenate([want0, t.step(9)])
        want_splats = t.get_splats().view(np.float32)
    res = []
    for rep in range(2):
        with S2D.MultiTrainer(268, 213, 1500, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
            m.set_target(tgt)
            m.init()
            got0 = m.step(1)
            img = m.get_image()
            got = np.concatenate([got0, m.step(4), m.step(5)])
            sp = m.get_splats()
            ad, b1, b2, it = m.get_adam()
            info = m.exchange_info()
            res.append((got.tobytes(), sp.tobytes(), ad.tobytes()))
        assert it == 10
        assert info["scheme"] == ("replicated" if replicated else "ownership")
        if not replicated:   # shared splats exist, and no rank holds everything
            assert info["rows_per_iteration"] > 0 and 1500 < info["held"] < 1500 * world
        assert img.tobytes() == want_img.tobytes()
        assert abs(got[0] - want[0]) <= 1e-12 * want[0]
        np.testing.assert_allclose(got, want, rtol=2e-5)
        np.testing.assert_allclose(sp.view(np.float32), want_splats, rtol=1e-3, atol=1e-3)
    assert res[0] == res[1]
    with pytest.raises(S2D.S2DError):
        S2D.MultiTrainer(268, 213, 100, [0, 0], replicated=replicated)   # two ranks on one GPU without the rehearsal flag


@pytest.mark.parametrize("world", [2, 4])
def test_multi_device_handle_ownership_equals_replicated_state_bit_for_bit(world):
    """The handle's two schemes form the same additions when gradients are deterministic and sums go in rank order
    (slab ownership: s2d_grads_combine; replicated state on a shared GPU: the host-staged sum), so whole runs are
    bit-identical: MSE trace, parameters and Adam moments after 150 iterations on a scene where splats change hands at
    the refreshes (every 64 iterations), and after a further stretch that follows a set_splats of the gathered state
    (hold sets made afresh from complete replicas)."""
    out = {}
    for replicated in (False, True):
        with S2D.MultiTrainer(768, 1024, 40000, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
            m.set_target_synthetic()
            m.init()
            tr = np.concatenate([m.step(50), m.step(100)])
            info = m.exchange_info()
            sp = m.get_splats()
            ad, b1, b2, it = m.get_adam()
            m.set_splats(sp)              # same values: the run must go on as if nothing had happened
            tr2 = m.step(30)
            sp2 = m.get_splats()
            ad2 = m.get_adam()[0]
            out[replicated] = (tr.tobytes(), sp.tobytes(), ad.tobytes(), tr2.tobytes(), sp2.tobytes(), ad2.tobytes())
            if not replicated:
                assert info["state_handovers"] > 0 and info["held"] < 40000 * world, info
            assert it == 150 and np.isfinite(tr).all() and tr[-1] < tr[0]
    names = ["trace", "splats", "adam", "trace after set_splats", "splats after", "adam after"]
    for k, name in enumerate(names):
        assert out[False][k] == out[True][k], name


@pytest.mark.parametrize("seed", range(8))
def test_multi_device_handle_random_sweep(seed):
    """Random rank counts (2..6 on this one GPU), image sizes and splat counts, 140+ iterations in uneven calls across two
    hold-set refreshes, with the Adam state read back and written again in the middle (which empties and refills the ranks'
    compact copies of their held splats): slab ownership and replicated state agree bit for bit, and both follow the single
    context's trace."""
    rng = np.random.default_rng(7000 + seed)
    world = int(rng.integers(2, 7))
    W = int(rng.integers(64, 500))
    H = int(rng.integers(16 * world, 16 * world + 500))
    n = int(rng.choice([300, 2000, 8000]))
    calls = [int(c) for c in rng.integers(1, 80, size=4)] + [70]
    tgt = O.synthetic_target(W, H)
    out = {}
    for replicated in (False, True):
        with S2D.MultiTrainer(W, H, n, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
            m.set_target(tgt)
            m.init()
            tr = []
            for k, c in enumerate(calls):
                tr.append(m.step(c))
                if k == 1:
                    ad, b1, b2, it = m.get_adam()
                    m.set_adam(ad, b1, b2, it)
                if k == 2:
                    m.set_splats(m.get_splats())
            tr = np.concatenate(tr)
            out[replicated] = (tr.tobytes(), m.get_splats().tobytes(), m.get_adam()[0].tobytes(), tr)
    for k, name in enumerate(["trace", "splats", "adam"]):
        assert out[False][k] == out[True][k], (name, world, W, H, n, calls)
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.init()
        want = t.step(sum(calls))
    # (the single context sums gradients with float atomics in another order: the runs drift apart slowly)
    np.testing.assert_allclose(out[False][3][:60], want[:60], rtol=5e-4)
    np.testing.assert_allclose(out[False][3], want, rtol=2e-2)
    assert abs(out[False][3][0] - want[0]) <= 1e-9 * want[0]


@pytest.mark.parametrize("world", [2, 3])
def test_multi_device_handle_forward_after_new_splats_uses_fresh_hold_sets(world):
    """step -> set_splats(other values) -> forward on a slab-ownership handle: the hold sets of the stepped run say
    nothing about the new parameters, so s2d_multi_forward makes them afresh before it rasterises -- a splat that now
    reaches a rank's rows without having been in its old set would otherwise be missing from that slab.  The stitched
    image equals a single context's frame of the same splats bit for bit."""
    W, H, n = 268, 213, 1500
    tgt = mini_target()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.init()
        first = t.get_splats()
        # the new parameters: the init() splats in reverse order, so nearly every splat sits in other rows than before
        new = first[::-1].copy()
        t.set_splats(new)
        t.forward()
        want = t.get_image()
    with S2D.MultiTrainer(W, H, n, [0] * world, share_gpu=True) as m:
        m.set_target(tgt)
        m.init()
        m.step(70)                        # hold sets made, refreshed once at 64, every context now holds a subset
        assert m.exchange_info()["held"] < n * world
        m.set_splats(new)
        m.forward()
        got = m.get_image()
        assert got.tobytes() == want.tobytes()
        m.set_splats(first)               # ... and back, without a step in between
        m.forward()
        back = m.get_image()
    with S2D.Trainer(W, H, n) as t:
        t.set_target(tgt)
        t.init()
        t.forward()
        assert back.tobytes() == t.get_image().tobytes()


def test_dormant_splats_wake_up_when_their_state_is_written_from_outside():
    """The Adam kernel skips blocks of splats whose moments are all zero and whose gradients came back zero (hidden splats:
    the step would leave them bit for bit as they are) and remembers them as dormant.  Every outside write of parameters or
    moments must cancel that: a context that has trained (and marked its hidden splats) is handed new moments / new
    parameters and must then step exactly like a fresh context handed the same state (deterministic gradients: bitwise)."""
    W, H, n = 512, 512, 60000          # dense enough that the high indices are hidden: whole blocks go dormant
    with S2D.Trainer(W, H, n, deterministic=True) as a:
        a.set_target_synthetic()
        a.init()
        a.step(6)
        sp = a.get_splats()
        ad, b1, b2, it = a.get_adam()
        hidden = np.flatnonzero(~ad["mv"].reshape(n, -1).any(axis=1))
        assert hidden.size > 2000 and (np.diff(hidden) == 1).sum() > 1000   # zero moments, in long runs of indices
        assert np.array_equal(sp[hidden], a_init_splats(W, H, n)[hidden])        # ... and they never moved
        # 1. new moments for some hidden splats (as a checkpoint load would bring): they must move now
        ad2 = ad.copy()
        ad2["mv"][hidden[::7], 0, 0] = 0.25     # m of pos.x
        ad2["mv"][hidden[::7], 0, 1] = 0.01     # v of pos.x
        a.set_adam(ad2, b1, b2, it)
        a.step(1)
        got = a.get_splats()
        with S2D.Trainer(W, H, n, deterministic=True) as b:
            b.set_target_synthetic()
            b.set_splats(sp)
            b.set_adam(ad2, b1, b2, it)
            b.step(1)
            want = b.get_splats()
        assert got.tobytes() == want.tobytes()
        assert np.all(got["pos"][hidden[::7], 0] != sp["pos"][hidden[::7], 0])
        # 2. parameters outside the constraints for hidden splats (set_splats): the next step clamps them (main.cpp:741-749)
        sp3 = got.copy()
        sp3["sx"][hidden[3::11]] = 0.25
        a.set_splats(sp3)
        a.step(1)
        assert np.all(a.get_splats()["sx"][hidden[3::11]] == 1.0)


def a_init_splats(W, H, n):
    with S2D.Trainer(W, H, n) as t:
        t.init()
        return t.get_splats()


def test_get_stats_respects_the_callers_struct_size():
    """s2d_stats starts with the size the CALLER was compiled with: the library writes no more than that (a caller built
    against a shorter, older struct is not overrun) and refuses a size below the struct's first version."""
    import ctypes as C
    L = S2D.load_library()
    with S2D.Trainer(64, 48, 10) as t:
        t.set_target_synthetic()
        t.init()
        t.step(2)
        full = C.sizeof(S2D._Stats)
        buf = (C.c_uint8 * (full + 64))()
        for i in range(len(buf)):
            buf[i] = 0xAB
        st = C.cast(buf, C.POINTER(S2D._Stats))
        shorter = S2D._Stats.bwd_quadrant_execs.offset + 8            # the struct as ABI version 2 introduced it
        st.contents.struct_size = shorter
        assert L.s2d_get_stats(t._h, st) == 0
        assert st.contents.iterations == 2 and st.contents.rebins >= 1 and st.contents.struct_size == shorter
        assert all(b == 0xAB for b in bytes(buf)[shorter:])         # nothing behind the caller's struct was touched
        st.contents.struct_size = shorter - 8
        assert L.s2d_get_stats(t._h, st) == 1                      # S2D_E_INVALID
        st.contents.struct_size = full + 32                        # a NEWER caller: gets what this library knows, sized to it
        assert L.s2d_get_stats(t._h, st) == 0 and st.contents.struct_size == full


def test_backward_refuses_a_framebuffer_the_fused_launch_did_not_store():
    """s2d_forward_backward with S2D_FB_SKIP_IMAGE leaves an OLDER frame in image0: a following s2d_backward, which reads
    image0, must refuse (S2D_E_STATE) instead of differentiating against that frame; with the image stored it runs."""
    o, t = make_pair(mini_target(), 1024, 2)
    with t:
        t.forward()
        t.backward()
        t.adam_step()
        t.forward_backward(skip_image=True)
        with pytest.raises(S2D.S2DError) as ei:
            t.backward()
        assert ei.value.code == 5 and "s2d_forward" in str(ei.value)
        t.forward_backward(skip_image=False)
        t.backward()   # same parameters, image0 current: allowed (adds a second copy of the gradients)
        t.synchronize()


@pytest.mark.parametrize("replicated", [False, True])
def test_multi_device_handle_reports_nonfinite_like_the_single_context(replicated):
    tgt = mini_target()
    with S2D.MultiTrainer(268, 213, 300, [0, 0], share_gpu=True, replicated=replicated) as m:
        m.set_target(tgt)
        m.init()
        m.step(2)
        ad, b1, b2, it = m.get_adam()
        ad["mv"][5, 4, 0] = np.inf
        m.set_adam(ad, b1, b2, it)
        with pytest.raises(S2D.S2DError) as ei:
            m.step(3)
        assert ei.value.code == 3
        m.init()                      # usable again after a restart
        assert np.isfinite(m.step(2)).all()


@pytest.mark.parametrize("world,replicated", [(2, False), (3, False), (2, True)])
def test_multi_device_handle_reports_a_rank_that_stopped_answering(world, replicated):
    """The reference's only failure policy is abort() (main.cpp:752-785); the boundary turns failures into statuses, and
    "a rank stopped answering" is one of them: with a stall injected into one rank's worker thread (include/splat2d_test.h)
    the step returns S2D_E_STATE inside the stall limit instead of hanging, the message names the rank and where the run
    stood, the handle refuses further steps, and destroying it does not hang either.  A stall SHORTER than the limit is
    just a slow rank: the run completes with the bits of an undisturbed one."""
    import time
    tgt = mini_target()
    with S2D.MultiTrainer(268, 213, 1500, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
        m.set_target(tgt)
        m.init()
        want = np.concatenate([m.step(3), m.step(4)])
    with S2D.MultiTrainer(268, 213, 1500, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
        m.set_target(tgt)
        m.init()
        m.set_stall_timeout(2000)
        m.test_stall(world - 1, 1, 300)          # slow, not gone
        got = np.concatenate([m.step(3), m.step(4)])
        assert got.tobytes() == want.tobytes()
        m.set_stall_timeout(400)
        m.test_stall(world - 1, 9, -1)           # gone: until somebody declares it so
        t0 = time.perf_counter()
        with pytest.raises(S2D.S2DError) as ei:
            m.step(5)
        dt = time.perf_counter() - t0
        assert ei.value.code == 5, ei.value      # S2D_E_STATE
        msg = str(ei.value)
        assert ("rank %d" % (world - 1)) in msg and ("stopped answering" in msg or "did not reach the rendezvous" in msg), msg
        assert "iteration 9" in msg, msg
        assert dt < 10.0, dt
        with pytest.raises(S2D.S2DError) as ei2:
            m.step(1)
        assert ei2.value.code == 5 and "create a new one" in str(ei2.value)
        t0 = time.perf_counter()
    assert time.perf_counter() - t0 < 10.0       # close() came back


@pytest.mark.parametrize("world", [2, 4])
def test_multi_device_handle_free_running_exchange_with_ranks_at_different_speeds(world):
    """Between refreshes the rank threads of a slab-ownership handle do not meet: they order the gradient exchange with
    pairwise sequence counters, stream event waits and two send buffers (csrc/s2d_multi.hip exchange_grads).  Make the
    ranks run at different speeds -- one rank's thread sleeps a few milliseconds at several iterations, another at
    others -- across three refresh intervals: parameters, moments and trace must equal those of the replicated scheme
    (whose sums are formed in the same rank order) bit for bit."""
    out = {}
    for replicated in (False, True):
        with S2D.MultiTrainer(512, 768, 20000, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
            m.set_target_synthetic()
            m.init()
            tr = []
            it = 0
            for k in range(52):                    # 52 x 4 = 208 iterations: three refreshes
                if not replicated:
                    m.test_stall((k * 7) % world, it + (k % 4), 3 + k % 5)
                tr.append(m.step(4))
                it += 4
            tr = np.concatenate(tr)
            info = m.exchange_info()
            out[replicated] = (tr.tobytes(), m.get_splats().tobytes(), m.get_adam()[0].tobytes())
            if not replicated:
                assert info["rows_per_iteration"] > 0 and info["state_handovers"] > 0, info
            assert np.isfinite(tr).all() and tr[-1] < tr[0]
    for k, name in enumerate(["trace", "splats", "adam"]):
        assert out[False][k] == out[True][k], name


@pytest.mark.parametrize("exchange", ["halo", "dense"])
def test_cpp_host_run_control_works_on_the_multi_device_handle(tmp_path, exchange):
    """splat2d_train --gpus 3 --share-gpu with the options the single-GPU loop has: a checkpoint
    written by a 3-rank run (state gathered from the holders) continues under a 2-rank run and under one context
    along the trajectory of an uninterrupted single-context run; Restart starts the trace over; --out-image of the
    multi-device run is the single context's frame to within the fp32 summation order of the gradients."""
    import subprocess
    exe = S2D._build.build_host_program()
    multi = ["--share-gpu", "--exchange", exchange]
    ck = str(tmp_path / "ck.bin")
    base = [exe, "--image", MINI, "--splats", "1024"]
    a = subprocess.run(base + ["--iters", "70", "--batch", "35", "--save-checkpoint", ck, "--gpus", "3"] + multi,
                       capture_output=True, text=True, check=True).stdout.strip().splitlines()   # (a hold-set refresh at 64)
    b = subprocess.run(base + ["--iters", "6", "--load-checkpoint", ck, "--gpus", "2", "--out-image", str(tmp_path / "m.ppm")] + multi,
                       capture_output=True, text=True, check=True).stdout.strip().splitlines()
    c = subprocess.run(base + ["--iters", "6", "--load-checkpoint", ck, "--out-image", str(tmp_path / "s.ppm")],
                       capture_output=True, text=True, check=True).stdout.strip().splitlines()
    full = subprocess.run(base + ["--iters", "76", "--batch", "38"], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    val = lambda lines: [float(l.split("mse")[1]) for l in lines]
    assert a[0] == "0 itr, mse 5934.9042" and [int(l.split()[0]) for l in b] == list(range(70, 76)) == [int(l.split()[0]) for l in c]
    np.testing.assert_allclose(val(a + b), val(full), rtol=2e-4)    # 76 iterations apart in summation order
    np.testing.assert_allclose(val(b), val(c), rtol=2e-5)           # the same checkpoint, 6 iterations
    im_m = np.frombuffer(open(tmp_path / "m.ppm", "rb").read()[15:], dtype=np.uint8).astype(int)
    im_s = np.frombuffer(open(tmp_path / "s.ppm", "rb").read()[15:], dtype=np.uint8).astype(int)
    assert im_m.size == 268 * 213 * 3 and np.abs(im_m - im_s).max() <= 1 and (im_m != im_s).mean() < 0.01
    r = subprocess.run(base + ["--iters", "5", "--restart-at", "3", "--gpus", "2"] + multi, capture_output=True, text=True,
                       check=True).stdout.strip().splitlines()
    assert [int(l.split()[0]) for l in r] == [0, 1, 2, 0, 1] and r[3] == "0 itr, mse 5934.9042"
    np.testing.assert_allclose(val(r[3:]), val(r[:2]), rtol=2e-5)
