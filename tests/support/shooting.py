"""Single shooting over many steps (the reference's TranscriptionMethod::SINGLE_SHOOTING, optimization/ocp.hpp:382-400:
only the inputs are decision variables, every state is an expression of all inputs before it): the flywheel of
test/src/optimization/flywheel_ocp_test.cpp — x' = -x + u discretized exactly, 5 s, track r = 10, |u| <= 12 — built
through the backend-agnostic Model wrapper so that the oracle and the product get the same model.  The Hessian of
the cost is DENSE in the N inputs: the reference factors such a system with its dense branch
(interior_point.hpp:340-352, util/dense_regularized_ldlt.hpp); a column of L (N entries) does not fit a task of the
product's sparse plan from a few hundred steps on, and the product takes its dense plan too (ldlt_dense_kernels.h)."""
import math

from tests.support import model

R = 10.0
U_MAX = 12.0


def build(m: model.Model, N: int, T: float = 5.0, final_state_constraint: bool = False) -> model.NlpProblem:
    p = model.NlpProblem(m)
    dt = T / N
    A = math.exp(-dt)
    B = 1.0 - A
    U = [p.decision_variable() for _ in range(N)]
    x = m.constant(0.0)
    cost = m.constant(0.0)
    for k in range(N):
        U[k].set_value(0.0)
        p.bounds(-U_MAX, U[k], U_MAX)
        cost = cost + m.pow(R - x, 2)
        x = A * x + B * U[k]
    cost = cost + m.pow(R - x, 2)
    if final_state_constraint:
        p.eq(x, R)
    p.minimize(cost)
    return p


def rollout(u, T: float = 5.0):
    """the states the inputs give (plain floats)"""
    N = len(u)
    dt = T / N
    A = math.exp(-dt)
    B = 1.0 - A
    xs = [0.0]
    for k in range(N):
        xs.append(A * xs[-1] + B * float(u[k]))
    return xs
