"""Iteration callbacks through the C-ABI (slpx_problem_add_callback, include/slpx.h) — the
behaviour the reference tests in test/src/optimization/solver/exit_status_test.cpp:17-45 (add_callback /
clear_callbacks, problem.hpp:690-712): called once per iteration with the iterate, a true
return stops the solve with CALLBACK_REQUESTED_STOP.
"""
import numpy as np
import pytest

import sleipnir_amd as sa
from tests.support import models

pytestmark = pytest.mark.gpu

CALLBACK_REQUESTED_STOP = 1  # exit_status.hpp:17


def test_callback_runs_once_per_iteration_and_sees_the_iterate():
    p = models.flywheel(50, 0.005)
    n, m_e, m_i = p.dims
    seen = []

    def cb(info):
        seen.append(info["iteration"])
        assert info["x"].shape == (n,) and info["y"].shape == (m_e,)
        assert info["s"].shape == (m_i,) and info["z"].shape == (m_i,)
        assert np.isfinite(info["f"]) and (info["s"] > 0).all() and (info["z"] > 0).all()
        return False

    p.add_callback(cb)
    status, rep = p.solve()
    assert status == 0
    assert seen == list(range(rep["iterations"]))
    # the solve is the same one with and without a (passive) callback
    q = models.flywheel(50, 0.005)
    status_q, rep_q = q.solve()
    assert status_q == 0 and rep_q["iterations"] == rep["iterations"]
    np.testing.assert_allclose(p.get_x(), q.get_x(), rtol=1e-9, atol=1e-12)
    p.close()
    q.close()


def test_callback_can_stop_the_solve_and_can_be_cleared():
    p = models.cart_pole(100, 5.0 / 100)  # (N = 50 is one of the fragile horizons)
    x0 = p.get_x()
    p.add_callback(lambda info: info["iteration"] == 3)
    status, rep = p.solve()
    assert status == CALLBACK_REQUESTED_STOP and rep["iterations"] == 3
    p.clear_callbacks()
    p.set_x(x0)
    status, rep = p.solve()
    assert status == 0 and rep["iterations"] > 3
    p.close()


def test_callback_matrices_follow_the_static_patterns():
    p = models.flywheel(20, 0.005)
    sysh = p.system()  # borrowed: same system solve() runs on
    n, m_e, m_i = p.dims
    assert (sysh.info["n"], sysh.info["m_e"], sysh.info["m_i"]) == (n, m_e, m_i)
    cp, ri = sysh.pattern(1)
    nnz_ae = sysh.info["nnz_Ae"]
    assert cp[-1] == nnz_ae and (ri < m_e).all()
    grabbed = {}

    def cb(info):
        off = info["off"]
        grabbed["Ae"] = np.array([info["V"][off[4] + k] for k in range(nnz_ae)])
        grabbed["ce"] = np.array([info["V"][off[1] + k] for k in range(m_e)])
        grabbed["x"] = info["x"]
        return True  # first iteration is enough

    p.add_callback(cb)
    status, _ = p.solve()
    assert status == CALLBACK_REQUESTED_STOP
    # flywheel dynamics are linear: c_e(x) = A_e x + const, so A_e must reproduce differences of c_e
    Ae = np.zeros((m_e, n))
    for c in range(n):
        Ae[ri[cp[c]:cp[c + 1]], c] = grabbed["Ae"][cp[c]:cp[c + 1]]
    assert np.isfinite(Ae).all() and np.abs(Ae).sum() > 0
    assert np.linalg.matrix_rank(Ae) == m_e
    p.close()


def test_model_changes_after_a_solve_are_honoured(fresh, slpx):
    """compile() used to keep the first compiled system for good, so constraints, costs
    or variables added after a solve() — and changed values of parameters — were silently ignored.
    The reference rebuilds its evaluators in every solve() (problem.hpp:517-660)."""
    from tests.support import model

    m = model.Model(model.ProductBackend("gpu"))
    m.be.reset()
    P = model.NlpProblem
    p = P(m)
    x, y = p.decision_variable(1.0), p.decision_variable(1.0)
    a = m.variable(3.0)  # a free Variable that is not a decision variable: a parameter
    p.minimize(m.pow(x - a, 2) + m.pow(y - 1, 2))
    assert p.solve() == P.SUCCESS
    assert abs(x.value() - 3.0) <= 1e-6 and abs(y.value() - 1.0) <= 1e-6
    a.set_value(-2.0)               # parameter change: folded constants are stale
    assert p.solve() == P.SUCCESS
    assert abs(x.value() + 2.0) <= 1e-6
    p.le(x + y, -4.0)               # new constraint after two solves
    assert p.solve() == P.SUCCESS
    assert x.value() + y.value() <= -4.0 + 1e-6
    assert abs((x.value() + 2.0) - (y.value() - 1.0)) <= 1e-5   # projection onto x + y = -4 from (-2, 1)
