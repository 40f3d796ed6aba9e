#!/usr/bin/env python3
"""Generates tests/golden/*.npz — Newton-step fixtures derived INDEPENDENTLY of both
the oracle and the product (SURVEY.md §8c "independent cross-checks"):

  * derivatives by second-order FORWARD-mode Taylor arithmetic ("jets": value,
    dense gradient, dense Hessian; exact to rounding), not by a reverse sweep over
    an expression graph;
  * the KKT system as a dense symmetric matrix, solved by numpy.linalg.solve (LU with
    partial pivoting), inertia by numpy.linalg.eigvalsh — no LDLᵀ, no ordering.

Problems are the reference's benchmark models re-stated from
benchmarks/scalability/cart_pole/sleipnir.cpp:16-129 and flywheel/sleipnir.cpp:12-42
(variable order, constraint order, initial guess), the IPM quantities from
solver/interior_point.hpp:426-481, the scaling from util/problem_scaling.hpp:100-107,
the (δ, γ) ladder from util/sparse_regularized_ldlt.hpp:82-151.

Run from the repo root:  python tests/golden/make_fixtures.py
Needs numpy only; does not import the oracle, the product or /root/reference.
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from tests.support import cases  # noqa: E402  (seeded IPM states only)


class Jet:
    """f, ∇f (n), ∇²f (n×n) propagated forward."""

    __slots__ = ("v", "g", "h")

    def __init__(self, v, g, h):
        self.v, self.g, self.h = v, g, h

    @staticmethod
    def const(c, n):
        return Jet(float(c), np.zeros(n), np.zeros((n, n)))

    @staticmethod
    def var(value, i, n):
        g = np.zeros(n)
        g[i] = 1.0
        return Jet(float(value), g, np.zeros((n, n)))

    def _lift(self, o):
        return o if isinstance(o, Jet) else Jet.const(o, len(self.g))

    def __add__(self, o):
        o = self._lift(o)
        return Jet(self.v + o.v, self.g + o.g, self.h + o.h)

    __radd__ = __add__

    def __neg__(self):
        return Jet(-self.v, -self.g, -self.h)

    def __sub__(self, o):
        return self + (-self._lift(o))

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o)
        return Jet(self.v * o.v, self.v * o.g + o.v * self.g,
                   self.v * o.h + o.v * self.h + np.outer(self.g, o.g) + np.outer(o.g, self.g))

    __rmul__ = __mul__

    def unary(self, f, d1, d2):
        return Jet(f, d1 * self.g, d1 * self.h + d2 * np.outer(self.g, self.g))

    def recip(self):
        r = 1.0 / self.v
        return self.unary(r, -r * r, 2 * r * r * r)

    def __truediv__(self, o):
        return self * self._lift(o).recip()

    def __rtruediv__(self, o):
        return self._lift(o) * self.recip()


def jsin(x):
    return x.unary(math.sin(x.v), math.cos(x.v), -math.sin(x.v))


def jcos(x):
    return x.unary(math.cos(x.v), -math.sin(x.v), -math.cos(x.v))


# ---- models ---------------------------------------------------------------------

def cart_pole_dynamics(x, u):
    m_c, m_p, l, g = cases.CP_MC, cases.CP_MP, cases.CP_L, cases.CP_G
    theta, thetadot = x[1], x[3]
    m00 = m_c + m_p
    m01 = m_p * l * jcos(theta)
    m11 = m_p * l * l
    c01 = -m_p * l * thetadot * jsin(theta)
    r0 = -(c01 * thetadot) + u[0]          # τ_g − C q̇ + B u, row 0
    r1 = -m_p * g * l * jsin(theta)        # row 1 (C's second row is zero)
    det = m00 * m11 - m01 * m01            # Cramer on the symmetric 2×2
    return [x[2], x[3], (r0 * m11 - m01 * r1) / det, (m00 * r1 - m01 * r0) / det]


def rk4(f, x, u, dt):
    def axpy(a, kx):
        return [xi + a * ki for xi, ki in zip(x, kx)]
    k1 = f(x, u)
    k2 = f(axpy(0.5 * dt, k1), u)
    k3 = f(axpy(0.5 * dt, k2), u)
    k4 = f(axpy(dt, k3), u)
    return [xi + dt / 6.0 * (a + 2.0 * b + 2.0 * c + d) for xi, a, b, c, d in zip(x, k1, k2, k3, k4)]


def cart_pole(N, xval):
    """Returns (f, c_e list, c_i list) as jets at xval; x0 layout X(4×(N+1)) row-major | U(N)."""
    n = 5 * N + 4
    dt = 5.0 / N
    v = [Jet.var(xval[i], i, n) for i in range(n)]
    X = lambda r, k: v[r * (N + 1) + k]  # noqa: E731
    U = lambda k: v[4 * (N + 1) + k]     # noqa: E731
    x_initial = [0.0, 0.0, 0.0, 0.0]
    x_final = [1.0, math.pi, 0.0, 0.0]
    ce = [X(r, 0) - x_initial[r] for r in range(4)] + [X(r, N) - x_final[r] for r in range(4)]
    ci = [X(0, k) - 0.0 for k in range(N + 1)] + [2.0 - X(0, k) for k in range(N + 1)]
    ci += [U(k) - (-20.0) for k in range(N)] + [20.0 - U(k) for k in range(N)]
    for k in range(N):
        nxt = rk4(cart_pole_dynamics, [X(r, k) for r in range(4)], [U(k)], dt)
        ce += [X(r, k + 1) - nxt[r] for r in range(4)]
    f = Jet.const(0.0, n)
    for k in range(N):
        f = f + U(k) * U(k)
    return f, ce, ci


def cart_pole_x0(N):
    x = np.zeros(5 * N + 4)
    for k in range(N + 1):
        x[k] = 0.0 + (1.0 - 0.0) * k / N
        x[(N + 1) + k] = math.pi * k / N
    return x


def flywheel(N, xval):
    n = 2 * N + 1
    dt = 5.0 / N
    A, B = math.exp(-dt), 1.0 - math.exp(-dt)
    v = [Jet.var(xval[i], i, n) for i in range(n)]
    X = lambda k: v[k]          # noqa: E731
    U = lambda k: v[N + 1 + k]  # noqa: E731
    ce = [X(k + 1) - (A * X(k) + B * U(k)) for k in range(N)] + [X(0) - 0.0]
    ci = [U(k) - (-12.0) for k in range(N)] + [12.0 - U(k) for k in range(N)]
    f = Jet.const(0.0, n)
    for k in range(N + 1):
        f = f + (10.0 - X(k)) * (10.0 - X(k))
    return f, ce, ci


# ---- Newton-step quantities ---------------------------------------------------------

LADDER = [(0.0, 0.0)] + [(10.0 ** e, 1e-10) for e in range(-4, 3)]


def inertia(K):
    w = np.linalg.eigvalsh(K)
    # eigvalsh is backward stable: absolute error ~ eps·‖K‖; anything below that is "zero"
    tol = 8 * np.finfo(float).eps * len(w) * max(1.0, float(np.max(np.abs(w))))
    return np.array([(w > tol).sum(), (w < -tol).sum(), (np.abs(w) <= tol).sum()], dtype=np.int64)


def newton_fixture(model, N, x0, case):
    n = len(x0)
    f0, ce0, ci0 = model(N, x0)
    me, mi = len(ce0), len(ci0)
    # scaling at the initial guess (problem_scaling.hpp:100-107)
    with np.errstate(divide="ignore"):
        d_f = min(1.0, 100.0 / np.max(np.abs(f0.g))) if np.max(np.abs(f0.g)) > 0 else 1.0
        d_ce = np.array([min(1.0, 100.0 / np.max(np.abs(c.g))) for c in ce0])
        d_ci = np.array([min(1.0, 100.0 / np.max(np.abs(c.g))) for c in ci0])
    if case == "indefinite":
        # interior state with large multipliers: −Σ yⱼ∇²cₑⱼ dominates and the reduced
        # Hessian loses positive definiteness, so the (δ, γ) ladder has to climb
        x, s, y, z, mu = cases.newton_state("interior", x0, n, me, mi, d_f)
        y = 200.0 * y
    else:
        x, s, y, z, mu = cases.newton_state(case, x0, n, me, mi, d_f)
    f, ce, ci = model(N, x)
    g = d_f * f.g
    c_e = d_ce * np.array([c.v for c in ce])
    c_i = d_ci * np.array([c.v for c in ci])
    A_e = d_ce[:, None] * np.array([c.g for c in ce]).reshape(me, n)
    A_i = d_ci[:, None] * np.array([c.g for c in ci]).reshape(mi, n)
    H = d_f * f.h
    for j in range(me):
        H = H - y[j] * d_ce[j] * ce[j].h
    for j in range(mi):
        H = H - z[j] * d_ci[j] * ci[j].h
    sigma = z / s
    top_left = H + A_i.T @ (sigma[:, None] * A_i)
    K = np.block([[top_left, A_e.T], [A_e, np.zeros((me, me))]])
    rhs = np.concatenate([-g + A_e.T @ y + A_i.T @ (-sigma * c_i + mu / s + z), -c_e])
    ladder_inertia = []
    chosen = None
    for d, gm in LADDER:
        reg = np.concatenate([np.full(n, d), np.full(me, -gm)])
        ine = inertia(K + np.diag(reg))
        ladder_inertia.append(ine)
        if chosen is None and ine[0] == n and ine[1] == me and ine[2] == 0:
            chosen = (d, gm)
    assert chosen is not None, "no ladder entry gives the ideal inertia"
    reg = np.concatenate([np.full(n, chosen[0]), np.full(me, -chosen[1])])
    p = np.linalg.solve(K + np.diag(reg), rhs)
    p_x, p_y = p[:n], -p[n:]
    p_s = (c_i - s) + A_i @ p_x
    p_z = mu / s - z - sigma * p_s
    return dict(
        N=N, n=n, m_e=me, m_i=mi, x0=x0, x=x, s=s, y=y, z=z, mu=mu, d_f=d_f, d_ce=d_ce, d_ci=d_ci,
        f=d_f * f.v, g=g, c_e=c_e, c_i=c_i, A_e=A_e, A_i=A_i, H=H, lhs=K, rhs=rhs,
        ladder=np.array(LADDER), ladder_inertia=np.array(ladder_inertia), chosen=np.array(chosen),
        p=p, p_x=p_x, p_y=p_y, p_s=p_s, p_z=p_z,
    )


def main():
    # cart-pole needs n = 5N+4 >= m_e = 4N+8, i.e. N >= 4 (N = 2 has too few DOFs)
    both = ("step0", "interior")
    jobs = [("cart_pole", cart_pole, 4, cart_pole_x0(4), both),
            ("cart_pole", cart_pole, 6, cart_pole_x0(6), both),
            ("cart_pole", cart_pole, 8, cart_pole_x0(8), ("indefinite",)),  # ladder climbs to δ = 10
            ("flywheel", flywheel, 5, np.zeros(11), both)]
    for name, model, N, x0, case_list in jobs:
        for case in case_list:
            fx = newton_fixture(model, N, x0, case)
            out = HERE / f"{name}_N{N}_{case}.npz"
            np.savez_compressed(out, **fx)
            print(out.name, "n=%d m_e=%d m_i=%d" % (fx["n"], fx["m_e"], fx["m_i"]),
                  "chosen (δ,γ) =", tuple(fx["chosen"]), "inertia ladder", fx["ladder_inertia"].tolist()[:3])


if __name__ == "__main__":
    main()
