"""The Python mirror of the reference's binding (sleipnir_amd.autodiff / sleipnir_amd.optimization,
reached here under the reference's own module names through sleipnir_amd.compat.install()),
exercised the way the reference's Python tests exercise theirs — same calls, same known answers:

  python/test/autodiff/variable_matrix_test.py, variable_test.py, gradient_test.py,
  jacobian_test.py, hessian_test.py
  python/test/optimization/{trivial,linear,quadratic,nonlinear,flywheel,cart_pole,
  double_integrator,arm_on_elevator}_problem_test.py, decision_variable_test.py,
  constraints_test.py, multistart_test.py, flywheel_ocp_test.py, differential_drive_ocp_test.py

CPU tier: everything that only builds models.  GPU tier: derivative VALUES (they come out of the
compiled tape on the device) and solves."""
import math

import numpy as np
import pytest

import sleipnir_amd.compat

sleipnir_amd.compat.install()

from sleipnir import autodiff  # noqa: E402
from sleipnir.autodiff import ExpressionType, Gradient, Hessian, Jacobian, Variable, VariableMatrix  # noqa: E402
from sleipnir.optimization import (OCP, DynamicsType, EqualityConstraints, ExitStatus,  # noqa: E402
                                   InequalityConstraints, Problem, TimestepMethod, TranscriptionMethod, bounds,
                                   multistart)


@pytest.fixture(autouse=True)
def _fresh_graph(fresh):
    yield


# ---- variable_test.py, variable_matrix_test.py -----------------------------------------------

def test_variable_constructors_and_values():
    assert Variable().value() == 0.0
    assert Variable(1).value() == 1.0 and Variable(1.5).value() == 1.5
    a = Variable()
    a.set_value(2)
    assert a.value() == 2.0
    assert Variable().type() == ExpressionType.LINEAR and Variable(3.0).type() == ExpressionType.CONSTANT
    assert (a * a).type() == ExpressionType.QUADRATIC and autodiff.sin(a).type() == ExpressionType.NONLINEAR


def test_matrix_default_assignment_and_aliasing():  # variable_matrix_test.py:8-57
    mat = VariableMatrix()
    assert (mat.rows(), mat.cols(), mat.shape) == (0, 0, (0, 0))
    mat = VariableMatrix(2, 2)
    assert mat.shape == (2, 2) and (mat.value() == 0.0).all()
    for i, (r, c) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        mat[r, c].set_value(i + 1.0)
    assert (mat.value() == [[1.0, 2.0], [3.0, 4.0]]).all()

    A = VariableMatrix([[1.0, 2.0], [3.0, 4.0]])
    B = VariableMatrix([[5.0, 6.0], [7.0, 8.0]])
    A = B
    B[0, 0].set_value(2.0)
    assert (A.value() == [[2.0, 6.0], [7.0, 8.0]]).all()


def test_matrix_slicing():  # variable_matrix_test.py:60-152
    mat = VariableMatrix([[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [13, 14, 15, 16]])
    for i in range(16):
        assert mat[i] == i + 1
    s = mat[1:, 2:]
    assert s.shape == (3, 2)
    assert [s[i].value() for i in range(6)] == [7.0, 8.0, 11.0, 12.0, 15.0, 16.0]
    assert s[2, 1] == 16.0
    s = mat[-1:, -2:]
    assert s.shape == (1, 2) and s[0] == 15.0 and s[0, 1] == 16.0
    assert (mat[:, ::2].value() == [[1.0, 3.0], [5.0, 7.0], [9.0, 11.0], [13.0, 15.0]]).all()
    assert (mat[::-1, ::-2].value() == [[16.0, 14.0], [12.0, 10.0], [8.0, 6.0], [4.0, 2.0]]).all()
    assert mat[1:, -1].shape == (3, 1) and (mat[1:, -1].value() == [[8.0], [12.0], [16.0]]).all()
    assert (mat[1:, -2].value() == [[7.0], [11.0], [15.0]]).all()
    mat[::2, ::2] = np.array([[17.0, 18.0], [19.0, 20.0]])
    assert (mat.value() == [[17.0, 2.0, 18.0, 4.0], [5.0, 6.0, 7.0, 8.0], [19.0, 10.0, 20.0, 12.0],
                            [13.0, 14.0, 15.0, 16.0]]).all()
    with pytest.raises(IndexError):
        mat[0, 1, 2]


@pytest.mark.parametrize("outer,inner,values,expected", [
    ((slice(None, None, 2), slice(None, None, 1)), (slice(1, 3), slice(1, 4)), [[1, 2, 3], [4, 5, 6]],
     [[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 1, 2, 3, 0], [0, 0, 0, 0, 0], [0, 4, 5, 6, 0]]),
    ((slice(None, None, -2), slice(None, None, -1)), (slice(1, 3), slice(1, 4)), [[1, 2, 3], [4, 5, 6]],
     [[0, 6, 5, 4, 0], [0, 0, 0, 0, 0], [0, 3, 2, 1, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]),
    ((slice(None, None, 1), slice(None, None, 2)), (slice(1, 4), slice(1, 3)), [[1, 2], [3, 4], [5, 6]],
     [[0, 0, 0, 0, 0], [0, 0, 1, 0, 2], [0, 0, 3, 0, 4], [0, 0, 5, 0, 6], [0, 0, 0, 0, 0]]),
    ((slice(None, None, -1), slice(None, None, -2)), (slice(1, 4), slice(1, 3)), [[1, 2], [3, 4], [5, 6]],
     [[0, 0, 0, 0, 0], [6, 0, 5, 0, 0], [4, 0, 3, 0, 0], [2, 0, 1, 0, 0], [0, 0, 0, 0, 0]]),
])
def test_matrix_subslicing(outer, inner, values, expected):  # variable_matrix_test.py:155-225
    mat = VariableMatrix(5, 5)
    mat[outer][inner] = np.array(values)
    assert (mat.value() == np.array(expected)).all()


def test_matrix_compound_assignment():  # variable_matrix_test.py:228-275
    A1, A2 = VariableMatrix([[1]]), VariableMatrix([[1, 2], [3, 4]])
    B, b = VariableMatrix([[1, 2], [3, 4]]), 2
    for target in (lambda: A2, lambda: A2[:, :]):
        t = target()
        t += B
        assert (A2.value() == [[2, 4], [6, 8]]).all()
        t = target()
        t -= B
        assert (A2.value() == [[1, 2], [3, 4]]).all()
        t = target()
        t *= B
        assert (A2.value() == [[7, 10], [15, 22]]).all()
        A2.set_value(np.array([[1, 2], [3, 4]]))
    A1 += b
    assert (A1.value() == [[3]]).all()
    A1 -= b
    assert (A1.value() == [[1]]).all()
    A2 *= b
    assert (A2.value() == [[2, 4], [6, 8]]).all()
    A2 /= b
    assert (A2.value() == [[1, 2], [3, 4]]).all()
    A2[:1, :1] += b
    assert (A2.value() == [[3, 2], [3, 4]]).all()
    A2[:1, :1] -= b
    A2[:, :] *= b
    assert (A2.value() == [[2, 4], [6, 8]]).all()
    A2[:, :] /= b
    assert (A2.value() == [[1, 2], [3, 4]]).all()


def test_matrix_iteration_values_and_element_functions():
    A = VariableMatrix([[1, 2, 3], [4, 5, 6], [7, 8, 9]])
    assert [v.value() for v in A] == [float(i) for i in range(1, 10)]
    assert [v.value() for v in A[1:, 1:]] == [5.0, 6.0, 8.0, 9.0]
    assert len(A) == 3
    assert A.value(1, 2) == 6.0 and A.value(5) == 6.0 and A.T.value(2, 1) == 6.0
    assert (A.T.value() == A.value().T).all()
    # cwise_map (cwise_transform), variable_matrix_test.py: cwise_map
    B = A.cwise_map(lambda x: x * 2.0)
    assert (B.value() == 2.0 * A.value()).all()
    assert np.allclose(A[:1, :].cwise_map(autodiff.sin).value(), np.sin([[1.0, 2.0, 3.0]]))
    # cwise_reduce
    C = autodiff.cwise_reduce(A, B, lambda x, y: x * y)
    assert (C.value() == A.value() * B.value()).all()
    # statics
    assert (VariableMatrix.zero(2, 3).value() == np.zeros((2, 3))).all()
    assert (VariableMatrix.one(2, 3).value() == np.ones((2, 3))).all()
    assert (VariableMatrix.identity(3).value() == np.eye(3)).all()
    assert (VariableMatrix.constant(2, 2, 4.5).value() == np.full((2, 2), 4.5)).all()


def test_matrix_arithmetic_with_numpy_on_either_side():
    X = VariableMatrix([[1.0, 2.0], [3.0, 4.0]])
    M = np.array([[0.0, 1.0], [1.0, 0.0]])
    assert (M @ X).value().tolist() == [[3.0, 4.0], [1.0, 2.0]]
    assert (X @ M).value().tolist() == [[2.0, 1.0], [4.0, 3.0]]
    assert ((M + X).value() == M + X.value()).all() and ((M - X).value() == M - X.value()).all()
    assert ((X - M).value() == X.value() - M).all()
    assert ((2.0 * X).value() == 2.0 * X.value()).all() and ((X / 2.0).value() == X.value() / 2.0).all()
    assert ((-X).value() == -X.value()).all()
    U = VariableMatrix(2, 2)
    U.set_value(X.value())
    J = sum(U[:, k:k + 1].T @ U[:, k:k + 1] for k in range(2))  # the cart-pole test's cost idiom
    assert J.shape == (1, 1) and J.value(0, 0) == 30.0 and Variable(J).type() == ExpressionType.QUADRATIC


def test_matrix_exponential():  # variable_matrix_test.py: exp()
    assert VariableMatrix([[4.0]]).exp().value() == pytest.approx(np.array([[math.exp(4.0)]]), abs=1e-13)
    for a in ([[0.0, 1.0], [0.0, -0.5]], [[0.0, 1.0], [0.0, 10.0]], [[1.0, 10.0], [0.0, 0.0]], [[2.0, 3.0], [4.0, 5.0]]):
        A = VariableMatrix(a)
        assert np.allclose(A.exp().value() @ (-A).exp().value(), np.eye(2), atol=1e-12)
        assert np.allclose(A[:, :].exp().value() @ (-A[:, :]).exp().value(), np.eye(2), atol=1e-12)
    pascal = VariableMatrix.zero(7, 7)
    for col in range(6):
        pascal[col + 1, col] = col + 1
    expected = np.array([[math.comb(r, c) for c in range(7)] for r in range(7)], dtype=float)
    assert np.allclose(pascal.exp().value(), expected, atol=1e-13)


def test_block_and_solve_free_functions():  # variable_matrix_test.py: block(), solve()
    A = VariableMatrix([[1.0, 2.0], [3.0, 4.0]])
    B = VariableMatrix([[5.0], [6.0]])
    C = VariableMatrix([[7.0, 8.0, 9.0]])
    assert autodiff.block([[A, B], [C]]).value().tolist() == [[1.0, 2.0, 5.0], [3.0, 4.0, 6.0], [7.0, 8.0, 9.0]]
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 5):
        a, b = rng.normal(size=(n, n)) + n * np.eye(n), rng.normal(size=(n, 2))
        X = autodiff.solve(VariableMatrix(a), VariableMatrix(b))
        assert np.allclose(X.value(), np.linalg.solve(a, b), atol=1e-12), n


# ---- decision_variable_test.py, trivial_problem_test.py, constraints_test.py -----------------

def test_decision_variables():
    problem = Problem()
    x = problem.decision_variable()
    assert isinstance(x, Variable) and x.value() == 0.0
    x.set_value(2.0)
    assert x.value() == 2.0
    y = problem.decision_variable(2)
    assert y.shape == (2, 1) and y.value(0) == 0.0
    y[0].set_value(1.0)
    y[1].set_value(2.0)
    assert (y.value(0), y.value(1)) == (1.0, 2.0)
    z = problem.decision_variable(3, 2)
    z.set_value(np.array([[7.0, 8.0], [9.0, 10.0], [11.0, 12.0]]))
    z[:2, :1].set_value(np.array([[1.0], [1.0]]))
    assert (z.value() == [[1.0, 8.0], [1.0, 10.0], [11.0, 12.0]]).all()
    A = problem.symmetric_decision_variable(2)
    A[0, 0].set_value(1.0)
    A[1, 0].set_value(2.0)
    A[1, 1].set_value(3.0)
    assert (A.value() == [[1.0, 2.0], [2.0, 3.0]]).all()


def test_trivial_problems_need_no_device():  # trivial_problem_test.py
    problem = Problem()
    assert problem.cost_function_type() == ExpressionType.NONE
    assert problem.equality_constraint_type() == ExpressionType.NONE
    assert problem.inequality_constraint_type() == ExpressionType.NONE
    assert problem.solve(diagnostics=True) == ExitStatus.SUCCESS
    problem = Problem()
    X = problem.decision_variable(2, 3)
    X.set_value(np.ones((2, 3)))
    assert problem.solve() == ExitStatus.SUCCESS
    assert (X.value() == 1.0).all()
    with pytest.raises(KeyError):
        problem.solve(tolerence=1e-3)  # misspelt keyword: bind_problem.cpp:108-111
    # the C entry point itself (slpx_problem_solve) takes the same early exit
    import sleipnir_amd

    low = sleipnir_amd.Problem()
    low.decision_variable()
    assert low.solve()[0] == 0
    low.close()


@pytest.mark.parametrize("lhs,rhs", [(1.0, 1.0), (1.0, 2.0), (2.0, 1.0)])
def test_constraint_truth_values(lhs, rhs):  # constraints_test.py
    kinds = {
        "scalar": lambda v: v, "Variable": Variable, "VariableMatrix": lambda v: VariableMatrix([[v]]),
        "block": lambda v: VariableMatrix([[v]])[:1, :1], "ndarray": lambda v: np.array([[v]]),
    }
    plain = ("scalar", "ndarray")
    for ln, lf in kinds.items():
        for rn, rf in kinds.items():
            if ln in plain and rn in plain:
                continue
            l, r = lf(lhs), rf(rhs)
            assert bool(l == r) == (lhs == rhs), (ln, rn)
            assert bool(l < r) == (lhs <= rhs) and bool(l <= r) == (lhs <= rhs), (ln, rn)
            assert bool(l > r) == (lhs >= rhs) and bool(l >= r) == (lhs >= rhs), (ln, rn)


def test_constraint_concatenation_and_bounds():
    eq1, eq2 = Variable(1) == Variable(1), Variable(1) == Variable(2)
    eqs = EqualityConstraints([eq1, eq2])
    assert (len(eq1.constraints), len(eq2.constraints), len(eqs.constraints)) == (1, 1, 2)
    assert eqs.constraints[1].value() == eq2.constraints[0].value()
    assert bool(eq1) and not bool(eq2) and not bool(eqs)
    in1, in2 = Variable(2) < Variable(1), Variable(1) < Variable(2)
    ins = InequalityConstraints([in1, in2])
    assert len(ins.constraints) == 2 and not bool(in1) and bool(in2) and not bool(ins)
    X = VariableMatrix([[0.5, 1.5]])
    b = bounds(0.0, X, 1.0)
    assert len(b.constraints) == 4 and not bool(b) and bool(bounds(0.0, X, 2.0))
    with pytest.raises(TypeError):
        Problem().subject_to(True)


def test_models_classify_like_the_reference():
    """cart_pole_problem_test.py:50-67 in miniature + arm_on_elevator's sin() inequality"""
    problem = Problem()
    X, U = problem.decision_variable(2, 4), problem.decision_variable(1, 3)
    for k in range(3):
        problem.subject_to(X[:, k + 1:k + 2] == X[:, k:k + 1] + 0.1 * autodiff.block([[X[1:2, k:k + 1]],
                                                                                     [U[:, k:k + 1]]]))
    problem.subject_to(X[:, :1] == np.zeros((2, 1)))
    problem.subject_to(bounds(-1.0, U, 1.0))
    problem.minimize(sum(U[:, k:k + 1].T @ U[:, k:k + 1] for k in range(3)))
    assert problem.cost_function_type() == ExpressionType.QUADRATIC
    assert problem.equality_constraint_type() == ExpressionType.LINEAR
    assert problem.inequality_constraint_type() == ExpressionType.LINEAR
    problem.subject_to(X[:1, :] + X[1:2, :].cwise_map(autodiff.sin) <= 1.8)
    assert problem.inequality_constraint_type() == ExpressionType.NONLINEAR


def test_symbolic_derivatives_need_no_device():
    x = VariableMatrix(3)
    for i in range(3):
        x[i].set_value(i + 1)
    y = VariableMatrix(3)
    y[0], y[1], y[2] = x[0] * x[1], x[1] * x[2], x[0] * x[2]
    assert (Jacobian(y, x).get().value() == [[2.0, 1.0, 0.0], [0.0, 3.0, 2.0], [3.0, 0.0, 1.0]]).all()
    f = x[0] * x[0] * x[1] + autodiff.sin(x[2])
    assert np.allclose(Gradient(f, x).get().value().T, [[4.0, 1.0, math.cos(3.0)]])
    assert np.allclose(Hessian(f, x).get().value(), [[4.0, 2.0, 0.0], [2.0, 0.0, 0.0], [0.0, 0.0, -math.sin(3.0)]])


# ---- GPU tier ---------------------------------------------------------------------------------

@pytest.mark.gpu
def test_derivative_values_come_from_the_device():
    """jacobian_test.py (products, nested products), gradient_test.py, hessian_test.py (sum of
    squares, product of sines): value() is the compiled tape on the GPU, get() the gradient tree."""
    x = VariableMatrix(3)
    for i in range(3):
        x[i].set_value(i + 1)
    y = VariableMatrix(3)
    y[0], y[1], y[2] = x[0] * x[1], x[1] * x[2], x[0] * x[2]
    J = Jacobian(y, x)
    assert (J.value().toarray() == [[2.0, 1.0, 0.0], [0.0, 3.0, 2.0], [3.0, 0.0, 1.0]]).all()
    x[0].set_value(5.0)  # the same evaluator at new values
    assert (J.value().toarray() == [[2.0, 5.0, 0.0], [0.0, 3.0, 2.0], [3.0, 0.0, 5.0]]).all()
    assert (J.get().value() == J.value().toarray()).all()

    v = VariableMatrix(5)
    for i in range(5):
        v[i].set_value(i + 1)
    f = sum(v[i] * v[i] for i in range(5))
    assert (Gradient(f, v).value().toarray().T == [[2.0, 4.0, 6.0, 8.0, 10.0]]).all()
    assert (Hessian(f, v).value().toarray() == 2.0 * np.eye(5)).all()
    g = autodiff.sin(v[0]) * autodiff.sin(v[1])
    H = Hessian(g, v[:2, :]).value().toarray()
    s0, s1, c0, c1 = math.sin(1.0), math.sin(2.0), math.cos(1.0), math.cos(2.0)
    assert np.allclose(H, [[-s0 * s1, c0 * c1], [c0 * c1, -s0 * s1]], atol=1e-15)
    assert np.allclose(Hessian(g, v[:2, :]).get().value(), H, atol=1e-15)


@pytest.mark.gpu
def test_small_problems_of_the_reference_tests():
    # linear_problem_test.py: maximize
    problem = Problem()
    x, y = problem.decision_variable(), problem.decision_variable()
    x.set_value(1.0)
    y.set_value(1.0)
    problem.maximize(50 * x + 40 * y)
    problem.subject_to(x + 1.5 * y <= 750)
    problem.subject_to(2 * x + 3 * y <= 1500)
    problem.subject_to(2 * x + y <= 1000)
    problem.subject_to(x >= 0)
    problem.subject_to(y >= 0)
    assert (problem.cost_function_type(), problem.equality_constraint_type(),
            problem.inequality_constraint_type()) == (ExpressionType.LINEAR, ExpressionType.NONE, ExpressionType.LINEAR)
    assert problem.solve(diagnostics=True) == ExitStatus.SUCCESS
    assert x.value() == pytest.approx(375.0, abs=1e-6) and y.value() == pytest.approx(250.0, abs=1e-6)

    # quadratic_problem_test.py: equality constrained (SQP entry point), unconstrained (Newton)
    problem = Problem()
    x, y = problem.decision_variable(), problem.decision_variable()
    problem.maximize(x * y)
    problem.subject_to(x + 3 * y == 36)
    assert problem.solve() == ExitStatus.SUCCESS
    assert x.value() == pytest.approx(18.0, abs=1e-5) and y.value() == pytest.approx(6.0, abs=1e-5)
    problem = Problem()
    x = problem.decision_variable()
    x.set_value(2.0)
    problem.minimize(x * x - 6.0 * x)
    assert problem.solve() == ExitStatus.SUCCESS and x.value() == pytest.approx(3.0, abs=1e-6)

    # nonlinear_problem_test.py: Rosenbrock with a disk constraint, one of its starts
    problem = Problem()
    x, y = problem.decision_variable(), problem.decision_variable()
    x.set_value(-1.5)
    y.set_value(-1.5)
    problem.minimize(autodiff.pow(1 - x, 2) + 100 * autodiff.pow(y - autodiff.pow(x, 2), 2))
    problem.subject_to(autodiff.pow(x, 2) + autodiff.pow(y, 2) <= 2)
    assert problem.solve() == ExitStatus.SUCCESS
    assert x.value() == pytest.approx(1.0, abs=1e-3) and y.value() == pytest.approx(1.0, abs=1e-3)


@pytest.mark.gpu
def test_callbacks_see_the_iterates_and_can_stop_the_solve():  # nonlinear_problem_test / problem.hpp:690-712
    problem = Problem()
    x, y = problem.decision_variable(), problem.decision_variable()
    problem.minimize((x - 2) ** 2 + (y + 1) ** 2)
    problem.subject_to(x + y == 0.5)
    problem.subject_to(x >= 0)
    seen = []

    def watch(info):
        seen.append((info.iteration, info.x.copy(), info.A_e.toarray(), info.H.toarray(), info.g.toarray()))

    problem.add_callback(watch)
    assert problem.solve() == ExitStatus.SUCCESS
    assert [s[0] for s in seen] == list(range(len(seen))) and len(seen) >= 2
    assert (seen[0][2] == [[1.0, 1.0]]).all() and (seen[0][3] == 2.0 * np.eye(2)).all()
    assert np.allclose(seen[-1][4].T, [[2 * (seen[-1][1][0] - 2), 2 * (seen[-1][1][1] + 1)]])
    problem.clear_callbacks()
    problem.add_callback(lambda info: info.iteration >= 1)
    x.set_value(0.0)
    y.set_value(0.0)
    assert problem.solve() == ExitStatus.CALLBACK_REQUESTED_STOP


@pytest.mark.gpu
def test_spy_files(tmp_path, monkeypatch):
    """problem_spy_test.py: solve(spy=True) writes H.spy, A_e.spy, A_i.spy (util/spy.hpp: labels,
    shape, then per iteration the coordinates with the signs of the entries)."""
    import struct

    monkeypatch.chdir(tmp_path)
    problem = Problem()
    x, y = problem.decision_variable(), problem.decision_variable()
    x.set_value(20.0)
    y.set_value(20.0)
    problem.minimize(x ** 4 + y ** 4)
    problem.subject_to(x >= 1)
    problem.subject_to(x <= 10)
    problem.subject_to(y == 2)
    iterations = []
    problem.add_callback(lambda info: iterations.append(info.iteration))
    assert problem.solve(spy=True) == ExitStatus.SUCCESS
    assert x.value() == pytest.approx(1.0, abs=1e-8) and y.value() == pytest.approx(2.0, abs=1e-8)

    def records(name, title, rows, cols):
        data = (tmp_path / name).read_bytes()
        pos = 0

        def i32():
            nonlocal pos
            pos += 4
            return struct.unpack_from("<i", data, pos - 4)[0]

        def text():
            nonlocal pos
            n = i32()
            pos += n
            return data[pos - n:pos].decode()

        assert (text(), text(), text()) == (title, "Constraints" if name != "H.spy" else "Decision variables", "Decision variables")
        assert (i32(), i32()) == (rows, cols)
        out = []
        while pos < len(data):
            rec = []
            for _ in range(i32()):
                r, c = i32(), i32()
                rec.append((r, c, chr(data[pos])))
                pos += 1
            out.append(rec)
        return out

    assert records("H.spy", "Hessian", 2, 2) == [[(0, 0, "+"), (1, 1, "+")]] * len(iterations)
    assert records("A_e.spy", "Equality constraint Jacobian", 1, 2) == [[(0, 1, "+")]] * len(iterations)
    assert records("A_i.spy", "Inequality constraint Jacobian", 2, 2) == [[(0, 0, "+"), (1, 0, "-")]] * len(iterations)


@pytest.mark.gpu
def test_double_integrator_through_the_python_interface():
    """double_integrator_problem_test.py: the bang-coast-bang profile (1e-4 away from switches),
    end points to 1e-8."""
    T, dt, r = 3.5, 0.005, 2.0
    N = int(T / dt)
    problem = Problem()
    X, U = problem.decision_variable(2, N + 1), problem.decision_variable(1, N)
    for k in range(N):
        problem.subject_to(X[0, k + 1] == X[0, k] + X[1, k] * dt + 0.5 * U[0, k] * dt * dt)
        problem.subject_to(X[1, k + 1] == X[1, k] + U[0, k] * dt)
    problem.subject_to(X[:, :1] == np.array([[0.0], [0.0]]))
    problem.subject_to(X[:, N:N + 1] == np.array([[r], [0.0]]))
    problem.subject_to(bounds(-1, X[1:2, :], 1))
    problem.subject_to(bounds(-1, U, 1))
    problem.minimize(sum((r - X[0, k]) ** 2 for k in range(N + 1)))
    assert problem.cost_function_type() == ExpressionType.QUADRATIC
    assert problem.equality_constraint_type() == ExpressionType.LINEAR
    assert problem.inequality_constraint_type() == ExpressionType.LINEAR
    assert problem.solve() == ExitStatus.SUCCESS
    u = U.value()[0]
    for k in range(1, N - 1):
        if abs(u[k - 1] - u[k + 1]) >= 1.0 - 1e-2:
            assert -1.0 <= u[k] <= 1.0
        else:
            t = k * dt
            want = 1.0 if t < 1.0 else 0.0 if t < 2.05 else -1.0 if t < 3.275 else 1.0
            assert u[k] == pytest.approx(want, abs=1e-4), k
    assert X.value(0, N) == pytest.approx(r, abs=1e-8) and X.value(1, N) == pytest.approx(0.0, abs=1e-8)


def _cart_pole_dynamics(x, u, fns):
    """test/include/cart_pole_util.hpp: M(q) q'' + C(q, q') q' = tau_g(q) + B u"""
    m_c, m_p, l, g = 5.0, 0.5, 0.5, 9.806
    sin, cos, mat, solve, stack = fns
    q, qdot, theta = x[:2, :], x[2:, :], x[1, 0]
    M = mat([[m_c + m_p, m_p * l * cos(theta)], [m_p * l * cos(theta), m_p * l ** 2]])
    C = mat([[0.0, -m_p * l * qdot[1, 0] * sin(theta)], [0.0, 0.0]])
    tau_g = mat([[0.0], [-m_p * g * l * sin(theta)]])
    B = np.array([[1.0], [0.0]])
    return stack(qdot, solve(M, tau_g + B @ u - C @ qdot))


def _rk4(f, x, u, dt):
    k1 = f(x, u)
    k2 = f(x + k1 * (dt / 2), u)
    k3 = f(x + k2 * (dt / 2), u)
    k4 = f(x + k3 * dt, u)
    return x + (k1 + k2 * 2.0 + k3 * 2.0 + k4) * (dt / 6.0)


@pytest.mark.gpu
def test_cart_pole_through_the_python_interface():
    """cart_pole_problem_test.py:19-113: N = 100 steps of 50 ms, SUCCESS, every state transition
    to 1e-8 of the RK4 model evaluated on plain numbers."""
    T, dt, u_max, d_max = 5.0, 0.05, 20.0, 2.0
    N = int(T / dt)
    x_final = np.array([[1.0], [math.pi], [0.0], [0.0]])
    sym = (autodiff.sin, autodiff.cos, VariableMatrix, autodiff.solve, lambda a, b: autodiff.block([[a], [b]]))
    num = (math.sin, math.cos, np.array, np.linalg.solve, lambda a, b: np.vstack([a, b]))

    problem = Problem()
    X = problem.decision_variable(4, N + 1)
    for k in range(N + 1):
        X[0, k].set_value(k / N * x_final[0, 0])
        X[1, k].set_value(k / N * x_final[1, 0])
    U = problem.decision_variable(1, N)
    problem.subject_to(X[:, :1] == np.zeros((4, 1)))
    problem.subject_to(X[:, N:N + 1] == x_final)
    problem.subject_to(bounds(0.0, X[:1, :], d_max))
    problem.subject_to(bounds(-u_max, U, u_max))
    for k in range(N):
        problem.subject_to(X[:, k + 1:k + 2] == _rk4(lambda x, u: _cart_pole_dynamics(x, u, sym), X[:, k:k + 1],
                                                     U[:, k:k + 1], dt))
    problem.minimize(sum(U[:, k:k + 1].T @ U[:, k:k + 1] for k in range(N)))
    assert problem.cost_function_type() == ExpressionType.QUADRATIC
    assert problem.equality_constraint_type() == ExpressionType.NONLINEAR
    assert problem.inequality_constraint_type() == ExpressionType.LINEAR
    assert problem.solve() == ExitStatus.SUCCESS
    Xv, Uv = X.value(), U.value()
    assert np.abs(Xv[:, 0]).max() < 1e-8 and np.abs(Xv[:, N:N + 1] - x_final).max() < 1e-8
    for k in range(N):
        assert X[0, k] >= -1e-12 and X[0, k] <= d_max + 1e-12 and abs(Uv[0, k]) <= u_max + 1e-9
        want = _rk4(lambda x, u: _cart_pole_dynamics(x, u, num), Xv[:, k:k + 1], Uv[:, k:k + 1], dt)
        assert np.abs(Xv[:, k + 1:k + 2] - want).max() < 1e-8, k


@pytest.mark.gpu
def test_multistart_runs_its_solves_on_threads():  # multistart_test.py:12-41
    def solve(guess):
        problem = Problem()
        x, y = problem.decision_variable(2)
        x.set_value(guess[0])
        y.set_value(guess[1])
        J = (autodiff.sin(y) * autodiff.exp((1 - autodiff.cos(x)) ** 2) +
             autodiff.cos(x) * autodiff.exp((1 - autodiff.sin(y)) ** 2) + (x - y) ** 2)
        problem.minimize(J)
        problem.subject_to((x + 5) ** 2 + (y + 5) ** 2 < 25)
        return problem.solve(), J.value(), (x.value(), y.value())

    status, _cost, (x, y) = multistart(solve, [(-3, -8), (-3, -1.5)])
    assert status == ExitStatus.SUCCESS
    assert x == pytest.approx(-3.13024680, abs=1e-8) and y == pytest.approx(-1.58214218, abs=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,method,steps", [
    (DynamicsType.EXPLICIT_ODE, TranscriptionMethod.DIRECT_TRANSCRIPTION, 1000),
    (DynamicsType.DISCRETE, TranscriptionMethod.DIRECT_TRANSCRIPTION, 1000),
    (DynamicsType.EXPLICIT_ODE, TranscriptionMethod.DIRECT_COLLOCATION, 1000),
    (DynamicsType.EXPLICIT_ODE, TranscriptionMethod.SINGLE_SHOOTING, 50),
], ids=["ode-transcription", "discrete-transcription", "ode-collocation", "ode-single-shooting"])
def test_flywheel_ocp_through_the_python_interface(kind, method, steps):
    """flywheel_ocp_test.py:20-120: states within 1e-2 of the discrete model under the bang-then-
    hold input, final state 10 rad/s.  (Single shooting at 50 steps: its Hessian is dense in the
    inputs, which the reference hands to its dense solver — outside SURVEY.md §8.)"""
    dt, r = 0.005, 10.0
    A, B = -1.0, 1.0
    Ad = math.exp(A * dt)
    Bd = (1.0 - Ad) * B
    f = (lambda x, u: A * x + B * u) if kind == DynamicsType.EXPLICIT_ODE else (lambda x, u: Ad * x + Bd * u)
    problem = OCP(1, 1, dt, steps, f, kind, TimestepMethod.FIXED, method)
    problem.constrain_initial_state(0.0)
    problem.set_upper_input_bound(12)
    problem.set_lower_input_bound(-12)
    r_mat = np.full((1, steps + 1), r)
    problem.minimize((r_mat - problem.X()) @ (r_mat - problem.X()).T)
    assert problem.cost_function_type() == ExpressionType.QUADRATIC
    assert problem.inequality_constraint_type() == ExpressionType.LINEAR
    assert problem.solve() == ExitStatus.SUCCESS
    X, U = problem.X().value()[0], problem.U().value()[0]
    near = lambda expected, actual, tol: abs(expected - actual) < tol
    if steps != 1000:
        # off the reference test's grid its tolerances (tuned to 5 ms steps) do not apply: compare
        # with direct transcription of the same dynamics, which poses the same discrete problem
        twin = OCP(1, 1, dt, steps, f, kind, TimestepMethod.FIXED, TranscriptionMethod.DIRECT_TRANSCRIPTION)
        twin.constrain_initial_state(0.0)
        twin.set_upper_input_bound(12)
        twin.set_lower_input_bound(-12)
        twin.minimize((r_mat - twin.X()) @ (r_mat - twin.X()).T)
        assert twin.solve() == ExitStatus.SUCCESS
        assert np.abs(X - twin.X().value()[0]).max() <= 1e-5
        assert np.abs(U[:steps] - twin.U().value()[0][:steps]).max() <= 1e-3
        return
    u_ss = (1.0 - Ad) * r / Bd  # holds r: r = Ad r + Bd u
    x = 0.0
    assert near(0.0, X[0], 1e-8)
    for k in range(steps):
        assert near(x, X[k], 1e-2), k
        u = 12.0 if r - x > 1e-2 else u_ss  # full voltage until the reference is reached, then hold
        if 0 < k < steps - 1 and near(12.0, U[k - 1], 1e-2) and near(u_ss, U[k + 1], 1e-2):
            assert u_ss <= U[k] <= 12.0, k
        else:
            assert near(u, U[k], 2.0 if method == TranscriptionMethod.DIRECT_COLLOCATION else 2e-4), k
        x = Ad * x + Bd * u
    assert near(r, X[steps], 2e-6)


@pytest.mark.gpu
def test_arm_on_elevator_through_the_python_interface():
    """arm_on_elevator_problem_test.py: N = 800, one nonlinear inequality per sample (the end
    effector's height, through cwise_map(autodiff.sin)); SUCCESS."""
    N, T = 800, 4.0
    dt = T / N
    problem = Problem()
    elevator, elevator_accel = problem.decision_variable(2, N + 1), problem.decision_variable(1, N)
    arm, arm_accel = problem.decision_variable(2, N + 1), problem.decision_variable(1, N)
    for k in range(N):
        problem.subject_to(elevator[0, k + 1] == elevator[0, k] + elevator[1, k] * dt + 0.5 * elevator_accel[0, k] * dt ** 2)
        problem.subject_to(elevator[1, k + 1] == elevator[1, k] + elevator_accel[0, k] * dt)
        problem.subject_to(arm[0, k + 1] == arm[0, k] + arm[1, k] * dt + 0.5 * arm_accel[0, k] * dt ** 2)
        problem.subject_to(arm[1, k + 1] == arm[1, k] + arm_accel[0, k] * dt)
    problem.subject_to(elevator[:, :1] == np.array([[1.0], [0.0]]))
    problem.subject_to(elevator[:, N:N + 1] == np.array([[1.25], [0.0]]))
    problem.subject_to(arm[:, :1] == np.array([[0.0], [0.0]]))
    problem.subject_to(arm[:, N:N + 1] == np.array([[math.pi], [0.0]]))
    problem.subject_to(bounds(-1.0, elevator[1:2, :], 1.0))
    problem.subject_to(bounds(-2.0, elevator_accel, 2.0))
    problem.subject_to(bounds(-2.0 * math.pi, arm[1:2, :], 2.0 * math.pi))
    problem.subject_to(bounds(-4.0 * math.pi, arm_accel, 4.0 * math.pi))
    heights = elevator[:1, :] + 1.0 * arm[:1, :].cwise_map(autodiff.sin)
    problem.subject_to(heights <= 1.8)
    problem.minimize(sum((1.25 - elevator[0, k]) ** 2 + (math.pi - arm[0, k]) ** 2 for k in range(N + 1)))
    assert problem.cost_function_type() == ExpressionType.QUADRATIC
    assert problem.equality_constraint_type() == ExpressionType.LINEAR
    assert problem.inequality_constraint_type() == ExpressionType.NONLINEAR
    assert problem.solve() == ExitStatus.SUCCESS
    assert (elevator.value()[0] + np.sin(arm.value()[0])).max() <= 1.8 + 1e-6


def _differential_drive(x, u):
    """test/include/differential_drive_util.hpp"""
    trackwidth, Kv_l, Ka_l, Kv_a, Ka_a = 0.699, 3.02, 0.642, 1.382, 0.08495
    A1, A2 = -(Kv_l / Ka_l + Kv_a / Ka_a) / 2.0, -(Kv_l / Ka_l - Kv_a / Ka_a) / 2.0
    B1, B2 = 0.5 / Ka_l + 0.5 / Ka_a, 0.5 / Ka_l - 0.5 / Ka_a
    v = (x[3, 0] + x[4, 0]) / 2.0
    xdot = VariableMatrix(5)
    xdot[0] = v * autodiff.cos(x[2, 0])
    xdot[1] = v * autodiff.sin(x[2, 0])
    xdot[2] = (x[4, 0] - x[3, 0]) / trackwidth
    xdot[3:5, :] = np.array([[A1, A2], [A2, A1]]) @ x[3:5, :] + np.array([[B1, B2], [B2, B1]]) @ u
    return xdot


@pytest.mark.gpu
def test_minimum_time_differential_drive_ocp_through_the_python_interface():
    """differential_drive_ocp_test.py: 50 steps, ONE shared timestep variable, minimum total
    time from the origin to (1, 1) at rest within +-12 V; SUCCESS, end points to 1e-8."""
    N, min_dt = 50, 0.05
    problem = OCP(5, 2, min_dt, N, _differential_drive, DynamicsType.EXPLICIT_ODE, TimestepMethod.VARIABLE_SINGLE,
                  TranscriptionMethod.DIRECT_TRANSCRIPTION)
    for i in range(N + 1):
        problem.X()[0, i].set_value(i / (N + 1))
        problem.X()[1, i].set_value(i / (N + 1))
    x_final = np.array([[1.0], [1.0], [0.0], [0.0], [0.0]])
    problem.constrain_initial_state(np.zeros((5, 1)))
    problem.constrain_final_state(x_final)
    problem.set_lower_input_bound(np.array([[-12.0], [-12.0]]))
    problem.set_upper_input_bound(np.array([[12.0], [12.0]]))
    problem.set_min_timestep(min_dt)
    problem.set_max_timestep(3.0)
    problem.minimize(problem.dt() @ np.ones((N + 1, 1)))
    assert problem.cost_function_type() == ExpressionType.LINEAR
    assert problem.equality_constraint_type() == ExpressionType.NONLINEAR
    assert problem.inequality_constraint_type() == ExpressionType.LINEAR
    assert problem.solve() == ExitStatus.SUCCESS
    X = problem.X().value()
    assert np.abs(X[:, 0]).max() < 1e-8 and np.abs(X[:, N:N + 1] - x_final).max() < 1e-8
    assert min_dt - 1e-9 <= problem.dt().value(0, 0) <= 3.0
