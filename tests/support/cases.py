"""Shared test cases: the synthetic Newton-step inputs of SURVEY.md §8(d) and the
comparison helpers used by both the CPU (host-check) and GPU parity tests."""
from __future__ import annotations

import numpy as np
from tests.support import models

SEED = 20260928

# SLPX_* switches the whole suite was started under (profiles/switch_matrix.sh runs it once per switch).  A test
# that asserts WHICH path a system took (multifrontal, supernode widths, twin attempts) does so only on the
# default paths — under an outer switch the assertion would be about the switch, not about the product; what the
# path computes is checked against the oracle either way.
import os as _os
OUTER_SWITCHES = {k: v for k, v in _os.environ.items() if k.startswith("SLPX_") and k != "SLPX_TWIN_VERBOSE"}


def build_pair(kind: str, N: int, sa, oracle):
    """Builds the same benchmark problem in the product and in the oracle."""
    dt = 5.0 / N
    if kind == "cart_pole":
        return models.cart_pole(N, dt), oracle.OracleProblem.cart_pole(N, dt)
    if kind == "flywheel":
        return models.flywheel(N, dt), oracle.OracleProblem.flywheel(N, dt)
    raise ValueError(kind)


def newton_state(case: str, x0, n, m_e, m_i, d_f, seed=SEED):
    """step0: interior_point.hpp:74-79 initial iterate.  interior: seeded perturbation."""
    if case == "step0":
        return x0.copy(), np.ones(m_i), np.zeros(m_e), np.ones(m_i), 0.1 * d_f
    rng = np.random.default_rng(seed)
    x = x0 + 1e-2 * rng.uniform(-1, 1, n)
    s = np.exp(rng.uniform(-2, 2, m_i))
    z = np.exp(rng.uniform(-2, 2, m_i))
    y = rng.uniform(-1, 1, m_e)
    return x, s, y, z, 0.1 * d_f


# Cart-pole model constants (benchmarks/scalability/cart_pole/sleipnir.cpp:42-45,
# test/include/cart_pole_util.hpp): m_c, m_p, l, g
CP_MC, CP_MP, CP_L, CP_G = 5.0, 0.5, 0.5, 9.806


def cart_pole_dynamics(x, u):
    """Plain-float ẋ = f(x, u) (cart_pole_util.hpp:43-78), x = [x, θ, ẋ, θ̇], u = [f_x]."""
    theta, thetadot = x[1], x[3]
    M = np.array([[CP_MC + CP_MP, CP_MP * CP_L * np.cos(theta)],
                  [CP_MP * CP_L * np.cos(theta), CP_MP * CP_L ** 2]])
    C = np.array([[0.0, -CP_MP * CP_L * thetadot * np.sin(theta)], [0.0, 0.0]])
    tau_g = np.array([0.0, -CP_MP * CP_G * CP_L * np.sin(theta)])
    qdd = np.linalg.solve(M, tau_g - C @ x[2:4] + np.array([1.0, 0.0]) * u[0])
    return np.array([x[2], x[3], qdd[0], qdd[1]])


def cart_pole_rk4(x, u, dt):
    """benchmarks/rk4.hpp:14-23 with zero-order-hold input."""
    k1 = cart_pole_dynamics(x, u)
    k2 = cart_pole_dynamics(x + 0.5 * dt * k1, u)
    k3 = cart_pole_dynamics(x + 0.5 * dt * k2, u)
    k4 = cart_pole_dynamics(x + dt * k3, u)
    return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


def cart_pole_unpack(xvec, N):
    """Decision vector -> X (4 × N+1), U (1 × N): decision_variable(4, N+1) then
    decision_variable(1, N), both row-major (cart_pole/sleipnir.cpp:87,98)."""
    xvec = np.asarray(xvec)
    return xvec[: 4 * (N + 1)].reshape(4, N + 1), xvec[4 * (N + 1):].reshape(1, N)


def csc_to_dict(colptr, rowidx, val):
    d = {}
    for c in range(len(colptr) - 1):
        for p in range(colptr[c], colptr[c + 1]):
            d[(int(rowidx[p]), c)] = d.get((int(rowidx[p]), c), 0.0) + float(val[p])
    return d


def lower_csc_to_dense_sym(colptr, rowidx, val, n):
    a = np.zeros((n, n))
    for c in range(n):
        for p in range(colptr[c], colptr[c + 1]):
            a[rowidx[p], c] += val[p]
            if rowidx[p] != c:
                a[c, rowidx[p]] += val[p]
    return a


def lower_csc_matvec(colptr, rowidx, val, x):
    """y = K x for a symmetric K given by its lower triangle."""
    y = np.zeros_like(x)
    for c in range(len(colptr) - 1):
        for p in range(colptr[c], colptr[c + 1]):
            r = rowidx[p]
            y[r] += val[p] * x[c]
            if r != c:
                y[c] += val[p] * x[r]
    return y


def regularized(colptr, rowidx, val, n, delta, gamma):
    """lhs + [δI 0; 0 −γI] on a lower CSC with a full diagonal."""
    out = np.array(val, dtype=np.float64, copy=True)
    for c in range(len(colptr) - 1):
        for p in range(colptr[c], colptr[c + 1]):
            if rowidx[p] == c:
                out[p] += delta if c < n else -gamma
    return out


def max_rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def sym_from_lower(colptr, rowidx, val):
    """scipy CSC of the full symmetric matrix given by its lower triangle."""
    import scipy.sparse as sp

    n = len(colptr) - 1
    low = sp.csc_matrix((np.asarray(val, dtype=np.float64), np.asarray(rowidx), np.asarray(colptr)), shape=(n, n))
    return (low + sp.tril(low, -1).T).tocsc()


def cond_inf_estimate(colptr, rowidx, val):
    """kappa_inf(K) = |K|_inf |K^-1|_inf for symmetric K (lower CSC): |K^-1|_1 by Hager/Higham's
    estimator over a sparse LU of K (scipy) — independent of both implementations under test."""
    import scipy.sparse.linalg as spla

    K = sym_from_lower(colptr, rowidx, val)
    lu = spla.splu(K)
    inv = spla.LinearOperator(K.shape, matvec=lu.solve, rmatvec=lambda b: lu.solve(b, trans="T"), dtype=np.float64)
    inv_norm = spla.onenormest(inv)  # symmetric: |.|_1 = |.|_inf
    return float(abs(K).sum(axis=1).max()) * float(inv_norm)


def refined_solution(colptr, rowidx, val, rhs, steps=3):
    """K x = rhs for symmetric K (lower CSC) to (nearly) full double accuracy whatever the
    conditioning: sparse LU with partial pivoting (scipy) + iterative refinement with residuals
    accumulated in extended precision (numpy longdouble).  The yardstick both the product's and
    the oracle's step are measured against."""
    import scipy.sparse.linalg as spla

    K = sym_from_lower(colptr, rowidx, val)
    lu = spla.splu(K)
    colptr, rowidx = np.asarray(colptr), np.asarray(rowidx)
    cols = np.repeat(np.arange(len(colptr) - 1), np.diff(colptr))
    v = np.asarray(val, dtype=np.longdouble)
    off = rowidx != cols
    b = np.asarray(rhs, dtype=np.longdouble)

    def matvec(x):
        y = np.zeros(len(x), dtype=np.longdouble)
        np.add.at(y, rowidx, v * x[cols])
        np.add.at(y, cols[off], v[off] * x[rowidx[off]])
        return y

    x = np.asarray(lu.solve(np.asarray(rhs, dtype=np.float64)), dtype=np.longdouble)
    for _ in range(steps):
        r = b - matvec(x)
        x = x + np.asarray(lu.solve(np.asarray(r, dtype=np.float64)), dtype=np.longdouble)
    return np.asarray(x, dtype=np.float64)
