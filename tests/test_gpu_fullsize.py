"""Full-size GPU tests at BASELINE.json's configurations, through size-independent properties (the oracle
needs ~20 s per iteration at 4096x4096 / 1 M Gaussians, so it checks a sub-sampled set of tiles here and the
small configurations in test_gpu_parity.py):
  * determinism of the forward pass (bit-identical twice), .w == 1, finite;
  * every tile list ascending in splat index (== the reference's blend order) and consistent with its offsets;
  * row-slab contexts tile the image bit-exactly and their partial squared errors add up;
  * a window of the 4096x4096 framebuffer equals the oracle's bit for bit (oracle run on the splats that can
    reach the window);
  * training decreases the MSE, parameters stay inside the clamps.
"""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
S2D = importlib.import_module("2dgaussiansplatting_amd")
D = importlib.import_module("2dgaussiansplatting_amd.distributed")


@pytest.fixture(scope="module")
def big():
    t = S2D.Trainer(4096, 4096, 1_000_000)
    t.set_target_synthetic()
    t.init()
    yield t
    t.close()


def test_cfg4_forward_deterministic_and_sane(big):
    big.forward()
    a = big.get_image()
    big.forward()
    b = big.get_image()
    assert a.tobytes() == b.tobytes()
    assert np.all(a[..., 3] == 1.0) and np.isfinite(a).all()
    assert 0.3 < a[..., :3].mean() < 0.5  # grey splats at opacity 1 saturate to ~0.5 almost everywhere


def test_cfg4_tile_lists_sorted(big):
    big.forward()
    tx, ty, off, lst = big.tile_lists()
    assert (tx, ty) == (256, 256) and off[0] == 0 and off[-1] == len(lst)
    d = np.diff(lst.astype(np.int64))
    starts = off[1:-1][(off[1:-1] > 0) & (off[1:-1] < len(lst))]
    bad = np.nonzero(d <= 0)[0] + 1            # positions where the index does not increase ...
    assert np.all(np.isin(bad, starts))        # ... are exactly tile boundaries
    assert 12e6 < len(lst) < 22e6


def test_cfg4_window_matches_oracle_bitwise(big):
    """Oracle on the sub-problem of one 64x48 window: only splats whose 3-sigma box can reach it matter, and
    blend order among them is still index order, so the window must be bit-identical."""
    big.forward()
    img = big.get_image()
    s = big.get_splats()
    x0, y0, w, h = 1000, 2000, 64, 48
    reach = 3.0 * np.maximum(s["sx"], s["sy"]) + 2.0
    sel = (s["pos"][:, 0] > x0 - reach) & (s["pos"][:, 0] < x0 + w + reach) & \
          (s["pos"][:, 1] > y0 - reach) & (s["pos"][:, 1] < y0 + h + reach)
    sub = s[sel]
    o = O.OracleTrainer(O.synthetic_target(4096, 4096)[:1, :1].repeat(4096, 0).repeat(4096, 1) * 0, len(sub))  # target unused
    o.splats[:] = sub.view(O.SPLAT_DTYPE)
    want = o.forward(y0, y0 + h)
    assert img[y0:y0 + h, x0:x0 + w].tobytes() == want[y0:y0 + h, x0:x0 + w].tobytes()


def test_cfg4_row_slabs_tile_the_image(big):
    big.forward()
    big.backward()
    full_img, full_mse = big.get_image(), big.mse()
    s = big.get_splats()
    mse = 0.0
    for rank in (0, 3):  # two of four slabs are enough to check placement and the partial errors
        r0, r1 = D.slab_rows(4096, rank, 4)
        with S2D.Trainer(4096, 4096, 1_000_000, row_begin=r0, row_end=r1) as t:
            t.set_target_synthetic()
            t.set_splats(s)
            t.forward()
            t.backward()
            part = t.get_image()
            assert part[r0:r1].tobytes() == full_img[r0:r1].tobytes()
            assert not part[:r0].any() and not part[r1:].any()
            assert t.stats()["pairs_binned"] < 0.3 * big.stats()["pairs_binned"]
            mse += t.mse()
    assert 0 < mse < full_mse


def test_cfg4_training_decreases_mse_and_respects_clamps():
    with S2D.Trainer(4096, 4096, 1_000_000) as t:
        t.set_target_synthetic()
        t.init()
        tr = t.step(30)
        s = t.get_splats()
    # Adam overshoots a little around iteration 8-12 (the reference's own trace plateaus there too), so the
    # decrease is asserted on a coarse grid
    assert np.all(np.diff(tr[:6]) < 0) and tr[10] < tr[0] / 3 and tr[29] < 0.2 * tr[0] and tr[29] < tr[15]
    assert s["pos"].min() >= 0 and s["pos"][:, 0].max() <= 4095 and s["pos"][:, 1].max() <= 4095
    assert s["sx"].min() >= 1 and s["sy"].min() >= 1 and s["color"].min() >= 0 and s["color"].max() <= 1
    assert np.all(s["opacity"] == 1.0)


def test_cfg3_2048_and_cfg5_8192_run():
    """BASELINE configs[2] (2048^2, 250 k) and the fp32 form of configs[4] (8192^2, 4 M): run, finite, sorted."""
    for W, n, iters in ((2048, 250_000, 5), (8192, 4_000_000, 2)):
        with S2D.Trainer(W, W, n) as t:
            t.set_target_synthetic()
            t.init()
            tr = t.step(iters)
            assert np.isfinite(tr).all() and tr[-1] < tr[0]
            st = t.stats()
            assert st["pairs_binned"] > 10 * n
