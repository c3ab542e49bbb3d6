"""End-to-end finite-difference check of the backward pass against the forward pass (row f4 of SURVEY.md section 8;
the reference's own validation method, main.cpp:51 + :642-701), independent of every known-answer vector:
  * CPU (always):  the oracle with exp_approx switched to expf -- its backward pass is the derivative of its forward pass;
  * GPU (-m gpu):  the HIP kernels with S2D_CFG_EXACT_EXP -- the same for the shipped forward / backward kernels, and
                   the two exact-exp implementations agree with each other.
With the reference's default exp_approx = (1 + x/8)^8 the analytic formulas are NOT the derivative of the forward pass
(they differentiate exp); test_default_approximation_is_not_the_derivative pins that down so the switch cannot rot
into a no-op."""
import importlib

import numpy as np
import pytest

import fd_check as FD
import oracle_lib as O


class OracleModel:
    def __init__(self, ref, n, exact):
        self.o = O.OracleTrainer(ref, n)
        self.exact = exact

    def _mode(self):
        O.lib().s2do_set_exact_exp(1 if self.exact else 0)

    def render(self, s9):
        self._mode()
        try:
            self.o.splats[:] = np.ascontiguousarray(s9).view(O.SPLAT_DTYPE).reshape(-1)
            return self.o.forward()
        finally:
            O.lib().s2do_set_exact_exp(0)

    def grads(self, s9):
        self._mode()
        try:
            self.o.splats[:] = np.ascontiguousarray(s9).view(O.SPLAT_DTYPE).reshape(-1)
            self.o.forward()
            return self.o.backward().view(np.float32).reshape(-1, 9).astype(np.float64)
        finally:
            O.lib().s2do_set_exact_exp(0)


def test_oracle_backward_is_the_derivative_of_its_forward_with_expf():
    s, ref = FD.scene()
    m = OracleModel(ref, len(s), exact=True)
    st = FD.check(m.render, m.grads(s), s, ref)
    print("\n[fd] oracle, expf: used %d skipped %d worst %.2e %s" % (st["used"], st["skipped"], st["worst_rel"], st["per_param"]))


def test_default_approximation_is_not_the_derivative():
    """(1 + x/8)^8 differs from exp(x) by up to 20 % inside the 3-sigma footprint and its slope by more: with the
    default exp_approx the same check must FAIL -- which is why the reference keeps the switch."""
    s, ref = FD.scene()
    m = OracleModel(ref, len(s), exact=False)
    with pytest.raises(AssertionError):
        FD.check(m.render, m.grads(s), s, ref)


def test_exact_exp_switch_leaves_the_default_path_alone():
    s, ref = FD.scene()
    a = OracleModel(ref, len(s), exact=False).render(s).copy()
    b = OracleModel(ref, len(s), exact=True).render(s).copy()
    c = OracleModel(ref, len(s), exact=False).render(s).copy()
    assert a.tobytes() == c.tobytes() and a.tobytes() != b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("deterministic", [False, True])
def test_hip_backward_is_the_derivative_of_the_hip_forward_with_expf(deterministic):
    S2D = importlib.import_module("2dgaussiansplatting_amd")
    s, ref = FD.scene()
    H, W = ref.shape[:2]
    with S2D.Trainer(W, H, len(s), exact_exp=True, deterministic=deterministic) as t:
        t.set_target(ref)

        def render(s9):
            t.set_splats(np.ascontiguousarray(s9).view(S2D.SPLAT_DTYPE).reshape(-1))
            t.forward()
            return t.get_image()

        render(s)
        t.backward()
        g = t.get_grads().view(np.float32).reshape(-1, 9).astype(np.float64)
        st = FD.check(render, g, s, ref)
    print("\n[fd] HIP, expf: used %d skipped %d worst %.2e %s" % (st["used"], st["skipped"], st["worst_rel"], st["per_param"]))
    # ... and the two expf implementations (ocml on the device, glibc on the host) agree to rounding
    m = OracleModel(ref, len(s), exact=True)
    want_img = m.render(s)
    with S2D.Trainer(W, H, len(s), exact_exp=True) as t:
        t.set_target(ref)
        t.set_splats(np.ascontiguousarray(s).view(S2D.SPLAT_DTYPE).reshape(-1))
        t.forward()
        img = t.get_image()
    assert np.abs(img - want_img).max() <= 2e-6
    w = m.grads(s)
    assert (np.abs(g - w) / np.maximum(np.abs(w), 1e-3 * np.abs(w).max(axis=0))).max() <= 1e-3


@pytest.mark.gpu
def test_hip_exact_exp_rejects_unsupported_combinations():
    S2D = importlib.import_module("2dgaussiansplatting_amd")
    for kw in ({"count_pairs": True}, {"fp16_images": True}):
        with pytest.raises(S2D.S2DError):
            S2D.Trainer(64, 64, 4, exact_exp=True, **kw)
