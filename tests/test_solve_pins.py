"""Whole-solve known answers from the reference's own optimization tests, run against
(a) the ORACLE's restatement of Problem::solve + interior_point (CPU tier — this pins
the oracle's IPM, filter, SOC, barrier update, regularization loop and restoration to
the reference's end states) and (b) the PRODUCT (`-m gpu` tier: C-ABI
slpx_problem_solve, Newton steps on the device).

Sources (problems, initial guesses, expected values and tolerances taken from there):
  test/src/optimization/linear_problem_test.cpp:14-61
  test/src/optimization/quadratic_problem_test.cpp:18-185
  test/src/optimization/nonlinear_problem_test.cpp:19-200
  test/src/optimization/solver/exit_status_test.cpp:17-230
  test/src/optimization/cart_pole_problem_test.cpp:87-124
  test/src/optimization/flywheel_problem_test.cpp:70-122

Like the reference, both implementations send problems without constraints to newton()
(problem.hpp:335, solver/newton.hpp), problems with equality constraints only to sqp()
(:403, solver/sqp.hpp) and everything else to interior_point() (:512).
"""
import math

import numpy as np
import pytest

from tests.support import cases, model
from tests.support import models

NONE, CONSTANT, LINEAR, QUADRATIC, NONLINEAR = 0, 1, 2, 3, 4  # expression_type.hpp:17-28
P = model.NlpProblem


@pytest.fixture(params=["oracle", pytest.param("product_gpu", marks=pytest.mark.gpu)])
def m(request, fresh):
    be = model.OracleBackend() if request.param == "oracle" else model.ProductBackend("gpu")
    be.reset()
    mm = model.Model(be)
    mm.is_oracle = request.param == "oracle"
    return mm


def near(a, b, tol):
    return abs(a - b) <= tol


# ---- linear_problem_test.cpp -------------------------------------------------------

def test_linear_maximize(m):  # :14-40
    p = P(m)
    x, y = p.decision_variable(1), p.decision_variable(1)
    p.maximize(50 * x + 40 * y)
    p.le(x + 1.5 * y, 750)
    p.le(2 * x + 3 * y, 1500)
    p.le(2 * x + y, 1000)
    p.ge(x, 0)
    p.ge(y, 0)
    assert p.types() == (LINEAR, NONE, LINEAR)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 375, 1e-6)
    assert near(y.value(), 250, 1e-6)


def test_linear_free_variable(m):  # :42-61
    p = P(m)
    x0, x1 = p.decision_variable(1), p.decision_variable(2)
    p.eq(x0, 0)
    assert p.types() == (NONE, LINEAR, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x0.value(), 0, 1e-6)
    assert near(x1.value(), 2, 1e-6)


# ---- quadratic_problem_test.cpp ----------------------------------------------------

def test_quadratic_unconstrained_1d(m):  # :18-35
    p = P(m)
    x = p.decision_variable(2)
    p.minimize(x * x - 6 * x)
    assert p.types() == (QUADRATIC, NONE, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 3, 1e-6)


def test_quadratic_unconstrained_2d(m):  # :37-80
    p = P(m)
    x, y = p.decision_variable(1), p.decision_variable(2)
    p.minimize(x * x + y * y)
    assert p.types() == (QUADRATIC, NONE, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 0, 1e-6)
    assert near(y.value(), 0, 1e-6)


def test_quadratic_equality_constrained(m):  # :82-160
    p = P(m)
    x, y = p.decision_variable(), p.decision_variable()
    p.maximize(x * y)
    p.eq(x + 3 * y, 36)
    assert p.types() == (QUADRATIC, LINEAR, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 18, 1e-5)
    assert near(y.value(), 6, 1e-5)

    p = P(m)
    x0, x1 = p.decision_variable(1), p.decision_variable(2)
    p.minimize(x0 * x0 + x1 * x1)
    p.eq(x0, 3)
    p.eq(x1, 3)
    assert p.types() == (QUADRATIC, LINEAR, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x0.value(), 3, 1e-5)
    assert near(x1.value(), 3, 1e-5)


def test_quadratic_inequality_constrained_2d(m):  # :162-185
    p = P(m)
    x, y = p.decision_variable(5), p.decision_variable(5)
    p.minimize(x * x + y * 2 * y)
    p.ge(y, -x + 5)
    assert p.types() == (QUADRATIC, NONE, LINEAR)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 3 + 1.0 / 3.0, 1e-6)
    assert near(y.value(), 1 + 2.0 / 3.0, 1e-6)


# ---- nonlinear_problem_test.cpp ----------------------------------------------------

def test_nonlinear_quartic(m):  # :19-38
    p = P(m)
    x = p.decision_variable(20)
    p.minimize(m.pow(x, 4))
    p.ge(x, 1)
    assert p.types() == (NONLINEAR, NONE, LINEAR)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 1, 1e-6)


def _grid(lo, hi, step):  # test/include/range.hpp: lo, lo+step, ... < hi
    n = int(math.floor((hi - lo) / step + 1e-9))
    return [lo + i * step for i in range(n)]


def test_nonlinear_rosenbrock_cubic_and_line(m):  # :40-81 (full grid on the oracle, every 9th start on the GPU tier)
    p = P(m)
    x, y = p.decision_variable(), p.decision_variable()
    p.minimize(100 * m.pow(y - m.pow(x, 2), 2) + m.pow(1 - x, 2))
    p.ge(y, m.pow(x - 1, 3) + 1)
    p.le(y, -x + 2)
    assert p.types() == (NONLINEAR, NONE, NONLINEAR)
    stride = 1 if m.is_oracle else 9
    for x0 in _grid(-1.5, 1.5, 0.1)[::stride]:
        for y0 in _grid(-0.5, 2.5, 0.1)[::stride]:
            x.set_value(x0)
            y.set_value(y0)
            assert p.solve() == P.SUCCESS, (x0, y0)
            # local minimum (0, 0), global minimum (1, 1)
            assert near(x.value(), 0, 1e-2) or near(x.value(), 1, 1e-2), (x0, y0, x.value())
            assert near(y.value(), 0, 1e-2) or near(y.value(), 1, 1e-2), (x0, y0, y.value())


def test_nonlinear_rosenbrock_disk(m):  # :83-116 (full grid on the oracle, every 9th start on the GPU tier)
    p = P(m)
    x, y = p.decision_variable(), p.decision_variable()
    p.minimize(m.pow(1 - x, 2) + 100 * m.pow(y - m.pow(x, 2), 2))
    p.le(m.pow(x, 2) + m.pow(y, 2), 2)
    assert p.types() == (NONLINEAR, NONE, QUADRATIC)
    stride = 1 if m.is_oracle else 9
    for x0 in _grid(-1.5, 1.5, 0.1)[::stride]:
        for y0 in _grid(-1.5, 1.5, 0.1)[::stride]:
            x.set_value(x0)
            y.set_value(y0)
            assert p.solve() == P.SUCCESS, (x0, y0)
            assert near(x.value(), 1, 1e-3), (x0, y0, x.value())
            assert near(y.value(), 1, 1e-3), (x0, y0, y.value())


def test_nonlinear_min_2d_distance_linear_constraint(m):  # :118-140
    p = P(m)
    x, y = p.decision_variable(20), p.decision_variable(50)
    p.minimize(m.sqrt(x * x + y * y))
    p.eq(y, -x + 5)
    assert p.types() == (NONLINEAR, LINEAR, NONE)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 2.5, 1e-2)
    assert near(y.value(), 2.5, 1e-2)


def test_nonlinear_conflicting_bounds(m):  # :142-161
    p = P(m)
    x, y = p.decision_variable(), p.decision_variable()
    p.minimize(m.hypot(x, y))
    p.le(m.hypot(x, y), 1)
    p.bounds(0.5, x, -0.5)
    assert p.types() == (NONLINEAR, NONE, NONLINEAR)
    assert p.solve() == P.GLOBALLY_INFEASIBLE


def test_nonlinear_wachter_biegler(m):  # :163-200
    p = P(m)
    x, s1, s2 = p.decision_variable(-2), p.decision_variable(3), p.decision_variable(1)
    p.minimize(x)
    p.eq(m.pow(x, 2) - s1 - 1, 0)
    p.eq(x - s2 - 0.5, 0)
    p.ge(s1, 0)
    p.ge(s2, 0)
    assert p.types() == (LINEAR, QUADRATIC, LINEAR)
    assert p.solve() == P.SUCCESS
    assert near(x.value(), 1, 1e-6)
    assert near(s1.value(), 0, 1e-6)
    assert near(s2.value(), 0.5, 1e-6)


# ---- multistart_test.cpp:17-53: Mishra's bird (single start nearest the optimum) ----

def test_mishras_bird(m):
    best = None
    for x0, y0 in [(-3.0, -8.0), (-3.0, -1.5)]:  # the two starts multistart() is given
        p = P(m)
        x, y = p.decision_variable(x0), p.decision_variable(y0)
        J = (m.sin(y) * m.exp(m.pow(1 - m.cos(x), 2)) + m.cos(x) * m.exp(m.pow(1 - m.sin(y), 2))
             + m.pow(x - y, 2))
        p.minimize(J)
        p.le(m.pow(x + 5, 2) + m.pow(y + 5, 2), 25)
        status = p.solve()
        xv, yv = x.value(), y.value()
        cost = (math.sin(yv) * math.exp((1 - math.cos(xv)) ** 2)
                + math.cos(xv) * math.exp((1 - math.sin(yv)) ** 2) + (xv - yv) ** 2)
        # multistart.hpp:64-79: prefer successful solves, then lowest cost
        key = (status != P.SUCCESS, cost)
        if best is None or key < best[0]:
            best = (key, status, xv, yv)
    _, status, xv, yv = best
    assert status == P.SUCCESS
    assert near(xv, -3.13024680, 1e-8)
    assert near(yv, -1.58214218, 1e-8)


# ---- exit_status_test.cpp ----------------------------------------------------------

def test_exit_too_few_dofs(m):  # :56-75
    p = P(m)
    x, y, z = p.decision_variables(3)
    p.eq(x, 1)
    p.eq(x, 2)
    p.eq(y, 1)
    p.eq(z, 1)
    assert p.types() == (NONE, LINEAR, NONE)
    assert p.solve() == P.TOO_FEW_DOFS


def test_exit_locally_infeasible(m):  # :77-117
    p = P(m)
    x, y, z = p.decision_variables(3)
    p.eq(x, y + 1)
    p.eq(y, z + 1)
    p.eq(z, x + 1)
    assert p.types() == (NONE, LINEAR, NONE)
    assert p.solve() == P.LOCALLY_INFEASIBLE

    p = P(m)
    x, y, z = p.decision_variables(3)
    p.ge(x, y + 1)
    p.ge(y, z + 1)
    p.ge(z, x + 1)
    assert p.types() == (NONE, NONE, LINEAR)
    assert p.solve() == P.LOCALLY_INFEASIBLE


@pytest.mark.parametrize("kind", ["cost_div", "cost_sqrt", "eq_div", "eq_sqrt", "ineq_div", "ineq_sqrt"])
def test_exit_nonfinite_initial_guess(m, kind):  # :119-170
    p = P(m)
    x = p.decision_variable()
    e = 1 / x if kind.endswith("div") else m.sqrt(x)
    if kind.startswith("cost"):
        p.minimize(e)
    elif kind.startswith("eq"):
        p.eq(e, 1)
    else:
        p.ge(e, 1)
    assert p.solve() == P.NONFINITE_INITIAL_GUESS


def test_exit_diverging_iterates(m):  # :172-187
    p = P(m)
    x = p.decision_variable()
    p.minimize(x)
    assert p.types() == (LINEAR, NONE, NONE)
    assert p.solve() == P.DIVERGING_ITERATES


def test_exit_max_iterations(m):  # :189-205
    p = P(m)
    x = p.decision_variable(1)
    p.minimize(x * x)
    assert p.types() == (QUADRATIC, NONE, NONE)
    assert p.solve(max_iterations=0) == P.MAX_ITERATIONS_EXCEEDED


def test_exit_timeout(m):  # :207-225 (timeout = 0 s there; both C-ABIs treat <= 0 as "none")
    p = P(m)
    x = p.decision_variable(1)
    p.minimize(x * x)
    assert p.solve(timeout=1e-12) == P.TIMEOUT


# ---- cart_pole_problem_test.cpp:87-124 / flywheel_problem_test.cpp:70-122 -----------

def _cart_pole_problem(m, N, dt):
    if m.is_oracle:
        from tests.support import oracle
        return oracle.OracleProblem.cart_pole(N, dt)
    return models.cart_pole(N, dt)


def _flywheel_problem(m, N, dt):
    if m.is_oracle:
        from tests.support import oracle
        return oracle.OracleProblem.flywheel(N, dt)
    return models.flywheel(N, dt)


def test_cart_pole_n100_end_state(m):
    N, T = 100, 5.0
    dt = T / N
    pr = _cart_pole_problem(m, N, dt)
    assert tuple(pr.types()) == (QUADRATIC, NONLINEAR, LINEAR)
    status, stats = pr.solve()
    assert status == P.SUCCESS
    X, U = cases.cart_pole_unpack(pr.get_x(), N)
    assert np.allclose(X[:, 0], [0, 0, 0, 0], atol=1e-8)
    assert np.allclose(X[:, N], [1, math.pi, 0, 0], atol=1e-8)
    assert np.all(X[0] >= -1e-9) and np.all(X[0] <= 2 + 1e-9)
    assert np.all(np.abs(U) <= 20 + 1e-9)
    for k in range(N):  # dynamics defects
        nxt = cases.cart_pole_rk4(X[:, k], U[:, k], dt)
        assert np.allclose(X[:, k + 1], nxt, atol=1e-8), k


def test_flywheel_end_state(m):
    # flywheel_problem_test.cpp: N = 5 s / 5 ms = 1000 steps on the oracle; the GPU tier
    # runs N = 200 to keep the suite short
    N = 1000 if m.is_oracle else 200
    dt = 0.005
    pr = _flywheel_problem(m, N, dt)
    assert tuple(pr.types()) == (QUADRATIC, LINEAR, LINEAR)
    status, stats = pr.solve()
    assert status == P.SUCCESS
    xs = pr.get_x()
    X, U = xs[: N + 1], xs[N + 1:]
    A, B = math.exp(-dt), 1 - math.exp(-dt)
    r = 10.0
    u_ss = 1.0 / B * (1.0 - A) * r  # :80-87
    assert near(X[0], 0.0, 1e-8)
    x, u = 0.0, 0.0
    for k in range(N):  # :93-118
        assert near(X[k], x, 1e-2), k
        u = 12.0 if r - x > 1e-2 else u_ss
        if 0 < k < N - 1 and near(12.0, U[k - 1], 1e-2) and near(u_ss, U[k + 1], 1e-2):
            assert u_ss <= U[k] <= 12.0, k
        else:
            assert near(U[k], u, 1e-4), k
        x = A * x + B * u
    if N == 1000:
        assert near(X[N], r, 2e-7)  # :121


# ---- solver dispatch (problem.hpp:335, 403, 512) --------------------------------------

@pytest.mark.gpu
def test_newton_and_sqp_iteration_counts_match_the_oracle():
    """Unconstrained -> newton(), equality-only -> sqp(): on these small, well-conditioned problems
    the product (device Newton steps) and the oracle take the same number of iterations — one for a
    quadratic without constraints (a single Newton step), and the same handful for the others."""
    def build(m):
        out = []
        p = P(m)                                      # quadratic_problem_test.cpp:37-80
        x, y = p.decision_variable(1.0), p.decision_variable(2.0)
        p.minimize(m.pow(x, 2) + m.pow(y, 2))
        out.append(("newton quadratic", p))
        p = P(m)                                      # nonlinear_problem_test.cpp:19-38
        x = p.decision_variable(20.0)
        p.minimize(m.pow(x - 1, 4))
        out.append(("newton quartic", p))
        p = P(m)                                      # quadratic_problem_test.cpp:82-160
        x, y = p.decision_variable(1.0), p.decision_variable(2.0)
        p.minimize(m.pow(x, 2) + m.pow(y, 2))
        p.eq(x + 3 * y, 36)
        out.append(("sqp quadratic", p))
        p = P(m)                                      # a nonlinear equality: needs several SQP steps
        x, y = p.decision_variable(2.0), p.decision_variable(1.0)
        p.minimize(m.pow(x - 2, 2) + m.pow(y - 1, 2))
        p.eq(m.pow(x, 2) + m.pow(y, 2), 2)
        out.append(("sqp circle", p))
        return out

    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("gpu"))
    mp.be.reset()
    for (name, po), (_, pp) in zip(build(mo), build(mp)):
        so, sp = po.solve(), pp.solve()
        io, ip = int(po.stats["iterations"]), int(pp.stats["iterations"])
        print(name, "status", so, sp, "iterations oracle", io, "product", ip)
        assert so == sp == P.SUCCESS
        assert io == ip, (name, io, ip)
        if name == "newton quadratic":
            assert ip == 1
