"""BASELINE config 5: the g-fold powered-descent guidance OCP, restated from the
reference's examples/g-fold/src/main.cpp:139-377 on the backend-agnostic modelling layer
(tests/support/model.py), so the SAME model is built in the oracle and in the product
through their expression C-ABIs.  Constants, variable order (X 6×(N+1), Z 1×(N+1), U 3×N,
σ 1×N, each row-major), constraint order and initial guesses follow the reference line by
line; A_d, B_d = blocks of expm([[A, B], [0, 0]]·dt) (main.cpp:30-48) via scipy.

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import expm

from . import model

M_WET, M_FUEL, T_MAX, ALPHA = 2000.0, 300.0, 24000.0, 5e-4
RHO_1, RHO_2 = 0.2 * T_MAX, 0.8 * T_MAX
Q_0, V_0 = np.array([2400.0, 450.0, -330.0]), np.array([-10.0, -40.0, 10.0])
Q_F, V_F = np.zeros(3), np.zeros(3)
G = np.array([-3.71, 0.0, 0.0])
OMEGA = np.array([2.53e-5, 0.0, 6.62e-5])
THETA = 90.0 * math.pi / 180.0
GAMMA_GS = 30.0 * math.pi / 180.0
V_MAX, DT = 90.0, 0.5


def discretized():
    w1, w2, w3 = OMEGA
    S = np.array([[0.0, -w3, w2], [w3, 0.0, -w1], [-w2, w1, 0.0]])
    A = np.zeros((6, 6))
    A[0:3, 3:6] = np.eye(3)
    A[3:6, 0:3] = -S @ S
    A[3:6, 3:6] = -2 * S
    B = np.zeros((6, 3))
    B[3:6, :] = np.eye(3)
    Mx = np.zeros((9, 9))
    Mx[0:6, 0:6], Mx[0:6, 6:9] = A, B
    phi = expm(Mx * DT)
    return phi[0:6, 0:6], phi[0:6, 6:9]


def n_range():
    """Horizon search interval (main.cpp:387-398)."""
    t_min = (M_WET - M_FUEL) * float(np.linalg.norm(V_0)) / RHO_2
    t_max = M_FUEL / (ALPHA * RHO_1)
    return math.ceil(t_min / DT), math.floor(t_max / DT)


def build(m: model.Model, N: int, end_straight: bool = True) -> model.NlpProblem:
    p = model.NlpProblem(m)
    A_d, B_d = discretized()
    X = [[p.decision_variable() for _ in range(N + 1)] for _ in range(6)]
    Z = [p.decision_variable() for _ in range(N + 1)]
    U = [[p.decision_variable() for _ in range(N)] for _ in range(3)]
    sig = [p.decision_variable() for _ in range(N)]

    def matvec(Mat, vec):  # variable_matrix.hpp:505-521: sum starts at 0 and accumulates
        out = []
        for i in range(Mat.shape[0]):
            acc = m.constant(0.0)
            for j in range(Mat.shape[1]):
                acc = acc + float(Mat[i, j]) * vec[j]
            out.append(acc)
        return out

    for i in range(3):
        p.eq(X[i][0], float(Q_0[i]))        # :241
    for i in range(3):
        p.eq(X[3 + i][0], float(V_0[i]))    # :244
    p.eq(Z[0], math.log(M_WET))             # :247
    for i in range(3):
        p.eq(X[i][N], float(Q_F[i]))        # :250
    for i in range(3):
        p.eq(X[3 + i][N], float(V_F[i]))    # :253
    for k in range(N + 1):                  # :256-263
        for i in range(3):
            X[i][k].set_value(Q_0[i] + (Q_F[i] - Q_0[i]) * k / N)
            X[3 + i][k].set_value(V_0[i] + (V_F[i] - V_0[i]) * k / N)

    for k in range(N + 1):                  # :266-373
        t = k * DT
        v_k = [X[3 + i][k] for i in range(3)]
        vtv = m.constant(0.0)
        for i in range(3):
            vtv = vtv + v_k[i] * v_k[i]
        p.le(vtv, V_MAX * V_MAX)            # :277
        z_min = math.log(M_WET - ALPHA * RHO_2 * t)
        z_max = math.log(M_WET - ALPHA * RHO_1 * t)
        z_est = (z_min + z_max) / 2
        Z[k].set_value(z_est)
        if k == N:
            continue
        x_k = [X[i][k] for i in range(6)]
        x_k1 = [X[i][k + 1] for i in range(6)]
        u_k = [U[i][k] for i in range(3)]
        s_k = sig[k]
        u_min, u_max = RHO_1 / math.exp(z_est), RHO_2 / math.exp(z_est)
        u_k[0].set_value((u_min + u_max) / 2)
        u_k[1].set_value(0.0)
        u_k[2].set_value(0.0)
        tg2 = math.tan(GAMMA_GS) * math.tan(GAMMA_GS)
        p.ge(m.pow(X[0][k] - float(Q_F[0]), 2),
             tg2 * (m.pow(X[1][k] - float(Q_F[1]), 2) + m.pow(X[2][k] - float(Q_F[2]), 2)))  # :319-322
        p.ge(s_k, 0.0)                      # :324
        if k == N - 1 and end_straight:
            p.eq(u_k[0], s_k)               # :334-336
            p.eq(u_k[1], 0.0)
            p.eq(u_k[2], 0.0)
        else:
            utu = m.constant(0.0)
            for i in range(3):
                utu = utu + u_k[i] * u_k[i]
            p.le(utu, s_k * s_k)            # :343
            p.ge(u_k[0], math.cos(THETA) * s_k)  # :352
        z_0 = math.log(M_WET - ALPHA * RHO_2 * t)
        mu_1, mu_2 = RHO_1 * math.exp(-z_0), RHO_2 * math.exp(-z_0)
        dz = Z[k] - z_0
        s_min = mu_1 * (1 - dz + 0.5 * m.pow(dz, 2))
        s_max = mu_2 * (1 - dz)
        p.bounds(s_min, s_k, s_max)         # :364
        s_k.set_value((s_min.value() + s_max.value()) / 2)
        rhs = matvec(A_d, x_k)
        gu = [float(G[i]) + u_k[i] for i in range(3)]
        bg = matvec(B_d, gu)
        for i in range(6):
            p.eq(x_k1[i], rhs[i] + bg[i])   # :375
        p.eq(Z[k + 1], Z[k] - ALPHA * DT * s_k)  # :376
    cost = m.constant(0.0)
    for s_k in sig:
        cost = cost + s_k                   # :380
    p.minimize(cost)
    return p


def dims(N: int, end_straight: bool = True):
    """(n, m_e, m_i) — SURVEY.md §8: n = 11N+7, m_e = 7N+16, m_i = 7N−1."""
    n = 6 * (N + 1) + (N + 1) + 3 * N + N
    m_e = 13 + 7 * N + (3 if end_straight else 0)
    m_i = (N + 1) + N * (1 + 1 + 2) + (N - 1 if end_straight else N) * 2
    return n, m_e, m_i
