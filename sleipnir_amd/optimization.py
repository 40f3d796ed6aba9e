"""The reference's Python interface `sleipnir.optimization`
(python/cpp/optimization/bind_problem.cpp:30-160, bind_ocp.cpp:25-130,
solver/bind_exit_status.cpp, solver/bind_iteration_info.cpp,
python/src/sleipnir/optimization/__init__.py:6-31) over the C-ABI of libslpx: Problem, OCP,
ExitStatus, bounds, the constraint types and multistart with the reference's names, argument
meaning and error behaviour.  solve() runs Problem::solve of the library (interior point, SQP
or Newton by the kinds of constraints present, problem.hpp:335,403,512) on the MI355X; there is
no CPU fallback — without a HIP device it raises.
"""
from __future__ import annotations

import concurrent.futures
import enum
import math

import numpy as np

import sleipnir_amd as _sa
from sleipnir_amd.autodiff import (EqualityConstraints, ExpressionType, InequalityConstraints, Variable,
                                   VariableMatrix, _is_scalar)

__all__ = ["ExitStatus", "Problem", "OCP", "DynamicsType", "TimestepMethod", "TranscriptionMethod",
           "EqualityConstraints", "InequalityConstraints", "IterationInfo", "bounds", "multistart"]


class ExitStatus(enum.IntEnum):
    """solver/exit_status.hpp:13-43"""
    SUCCESS = 0
    CALLBACK_REQUESTED_STOP = 1
    TOO_FEW_DOFS = -1
    LOCALLY_INFEASIBLE = -2
    GLOBALLY_INFEASIBLE = -3
    FACTORIZATION_FAILED = -4
    LINE_SEARCH_FAILED = -5
    FEASIBILITY_RESTORATION_FAILED = -6
    NONFINITE_INITIAL_GUESS = -7
    DIVERGING_ITERATES = -8
    MAX_ITERATIONS_EXCEEDED = -9
    TIMEOUT = -10


def bounds(l, x, u) -> InequalityConstraints:
    """variable.hpp:1008-1013: l <= x <= u"""
    return InequalityConstraints([l <= x, x <= u])


class IterationInfo:
    """solver/iteration_info.hpp:13-41: iteration, x, s, y, z as arrays; g, H, A_e, A_i as scipy
    matrices over the static patterns (built when asked for)."""

    def __init__(self, raw, problem):
        self.iteration = raw["iteration"]
        self.x, self.s, self.y, self.z = raw["x"], raw["s"], raw["y"], raw["z"]
        self.in_restoration = raw["in_restoration"]
        self._raw, self._problem = raw, problem
        # the value vector belongs to the solver: copy what the properties below need now
        self._V = None
        if not self.in_restoration:
            end = problem._value_count()
            if end:
                self._V = np.ctypeslib.as_array(raw["V"], shape=(end,)).copy()

    def _matrix(self, which, off_index, shape, lower=False):
        import scipy.sparse

        if self._V is None:
            raise RuntimeError("matrices are not available inside feasibility restoration")
        cp, ri = self._problem._patterns()[which]
        off = self._raw["off"][off_index]
        nnz = int(cp[-1])
        m = scipy.sparse.csc_matrix((self._V[off:off + nnz], np.asarray(ri[:nnz]), np.asarray(cp)), shape=shape)
        m.sum_duplicates()
        return m

    @property
    def g(self):
        n = len(self.x)
        return self._matrix(0, 3, (1, n)).T.tocsc()

    @property
    def A_e(self):
        return self._matrix(1, 4, (len(self.y), len(self.x)))

    @property
    def A_i(self):
        return self._matrix(2, 5, (len(self.z), len(self.x)))

    @property
    def H(self):
        """Hessian of the Lagrangian, lower triangle (iteration_info.hpp:34): H_f + H_c"""
        n = len(self.x)
        return (self._matrix(3, 6, (n, n)) + self._matrix(4, 7, (n, n))).tocsc()


class Problem:
    """problem.hpp:66-720"""

    def __init__(self):
        self._p = _sa.Problem()
        self._decision_variables: list[Variable] = []
        self._pattern_cache = None
        self.report = None  # counters and phase times of the last solve() that ran an iteration

    # ---- model ----
    def decision_variable(self, rows=None, cols=1):
        """problem.hpp:78-104: no argument -> Variable, (rows[, cols]) -> VariableMatrix"""
        if rows is None:
            return self._new_variable()
        out = np.empty((int(rows), int(cols)), dtype=object)
        for r in range(out.shape[0]):
            for c in range(out.shape[1]):
                out[r, c] = self._new_variable()
        return VariableMatrix._of(out)

    def symmetric_decision_variable(self, rows):
        """problem.hpp:118-140: only the lower triangle is new variables"""
        out = np.empty((int(rows), int(rows)), dtype=object)
        for r in range(out.shape[0]):
            for c in range(r + 1):
                out[r, c] = out[c, r] = self._new_variable()
        return VariableMatrix._of(out)

    def _new_variable(self) -> Variable:
        self._pattern_cache = None
        v = Variable._wrap(self._p.decision_variable())
        self._decision_variables.append(v)
        return v

    def minimize(self, cost):
        self._pattern_cache = None
        self._p.minimize(Variable._lift(cost).node)

    def maximize(self, objective):
        self._pattern_cache = None
        self._p.maximize(Variable._lift(objective).node)

    def subject_to(self, constraint):
        self._pattern_cache = None
        if isinstance(constraint, EqualityConstraints):
            for c in constraint.constraints:
                self._p.subject_to_eq(c.node)
        elif isinstance(constraint, InequalityConstraints):
            for c in constraint.constraints:
                self._p.subject_to_ineq(c.node)
        else:
            raise TypeError("subject_to() takes EqualityConstraints or InequalityConstraints "
                            "(the result of ==, <=, >= or bounds())")

    def cost_function_type(self): return ExpressionType(self._p.types()[0])
    def equality_constraint_type(self): return ExpressionType(self._p.types()[1])
    def inequality_constraint_type(self): return ExpressionType(self._p.types()[2])

    # ---- solve ----
    def solve(self, **kwargs) -> ExitStatus:
        """problem.hpp:281-679.  Keyword arguments: tolerance, max_iterations, timeout,
        feasible_ipm, diagnostics, spy (bind_problem.cpp:80-112); anything else is a KeyError like
        the reference's."""
        allowed = {"tolerance", "max_iterations", "timeout", "feasible_ipm", "diagnostics", "spy"}
        for k in kwargs:
            if k not in allowed:
                raise KeyError(f"Invalid keyword argument: {k}")
        timeout = kwargs.pop("timeout", 0.0)
        if timeout is None or math.isinf(timeout):
            timeout = 0.0
        # problem.hpp:304-313: nothing to do for a problem without cost and constraints
        if all(t <= ExpressionType.CONSTANT for t in self._p.types()):
            self.report = None
            return ExitStatus.SUCCESS
        status, self.report = self._p.solve(timeout=timeout, **kwargs)
        return ExitStatus(status)

    def add_callback(self, callback):
        """problem.hpp:690-709: callback(info) -> True to stop (None counts as False)"""
        self._p.add_callback(lambda raw: bool(callback(IterationInfo(raw, self))))

    def clear_callbacks(self):
        self._p.clear_callbacks()

    # ---- helpers of IterationInfo ----
    def _patterns(self):
        if self._pattern_cache is None:
            s = self._p.system()
            self._pattern_cache = ({k: s.pattern(k) for k in range(5)}, s.info)
        return self._pattern_cache[0]

    def _value_count(self):
        self._patterns()
        return int(self._pattern_cache[1].get("nV", 0))

    def close(self):
        if self._p is not None:
            self._p.close()
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DynamicsType(enum.IntEnum):
    """ocp/dynamics_type.hpp:9-14"""
    EXPLICIT_ODE = 0
    DISCRETE = 1


class TimestepMethod(enum.IntEnum):
    """ocp/timestep_method.hpp:9-17"""
    FIXED = 0
    VARIABLE = 1
    VARIABLE_SINGLE = 2


class TranscriptionMethod(enum.IntEnum):
    """ocp/transcription_method.hpp:9-19"""
    DIRECT_TRANSCRIPTION = 0
    DIRECT_COLLOCATION = 1
    SINGLE_SHOOTING = 2


class OCP(Problem):
    """ocp.hpp:49-400 as the Python binding shows it (bind_ocp.cpp:29-128): dynamics is f(x, u)
    on column VariableMatrix arguments, dt a number of seconds."""

    def __init__(self, num_states, num_inputs, dt, num_steps, dynamics, dynamics_type=DynamicsType.EXPLICIT_ODE,
                 timestep_method=TimestepMethod.FIXED, transcription_method=TranscriptionMethod.DIRECT_TRANSCRIPTION):
        super().__init__()
        self._samples = int(num_steps) + 1
        self._f = dynamics
        self._kind = DynamicsType(dynamics_type)
        N1 = self._samples
        # one input more than there are steps: the last sample's constraints need one too (ocp.hpp:122-123)
        self._U = self.decision_variable(num_inputs, N1)
        if timestep_method == TimestepMethod.FIXED:
            self._DT = VariableMatrix(np.full((1, N1), float(dt)))
        elif timestep_method == TimestepMethod.VARIABLE_SINGLE:
            shared = self.decision_variable()
            shared.set_value(dt)
            self._DT = VariableMatrix._of(np.full((1, N1), shared, dtype=object))
        else:
            self._DT = self.decision_variable(1, N1)
            for k in range(N1):
                self._DT[0, k].set_value(dt)
        if transcription_method == TranscriptionMethod.SINGLE_SHOOTING:
            self._X = VariableMatrix(num_states, N1)  # expressions, filled by the rollout
        else:
            self._X = self.decision_variable(num_states, N1)
        if transcription_method == TranscriptionMethod.DIRECT_COLLOCATION and self._kind != DynamicsType.EXPLICIT_ODE:
            raise ValueError("OCP: direct collocation needs an explicit ODE")  # slp_assert at ocp.hpp:323

        for k in range(N1 - 1):
            h = self._DT[0, k]
            x0, u0 = self._X[:, k:k + 1], self._U[:, k:k + 1]
            if transcription_method == TranscriptionMethod.DIRECT_TRANSCRIPTION:  # ocp.hpp:359-379
                self.subject_to(self._X[:, k + 1:k + 2] == self._step(x0, u0, h))
            elif transcription_method == TranscriptionMethod.SINGLE_SHOOTING:  # ocp.hpp:382-401
                self._X[:, k + 1:k + 2] = self._step(x0, u0, h)
            else:  # ocp.hpp:322-357 (Hermite-Simpson)
                x1, u1 = self._X[:, k + 1:k + 2], self._U[:, k + 1:k + 2]
                f0, f1 = self._f(x0, u0), self._f(x1, u1)
                xdot_mid = (-3.0 / (2.0 * h)) * (x0 - x1) - 0.25 * (f0 + f1)
                x_mid = 0.5 * (x0 + x1) + (h / 8.0) * (f0 - f1)
                u_mid = 0.5 * (u0 + u1)
                self.subject_to(xdot_mid == self._f(x_mid, u_mid))

    def _step(self, x, u, h):
        """x_{k+1} as an expression of (x_k, u_k, h): the transition function itself, or one
        classical Runge-Kutta step of the ODE (ocp.hpp:310-319)"""
        if self._kind == DynamicsType.DISCRETE:
            return self._f(x, u)
        half = h * 0.5
        k1 = self._f(x, u)
        k2 = self._f(x + k1 * half, u)
        k3 = self._f(x + k2 * half, u)
        k4 = self._f(x + k3 * h, u)
        return x + (k1 + k2 * 2.0 + k3 * 2.0 + k4) * (h / 6.0)

    @staticmethod
    def _column(v, rows):
        if _is_scalar(v):
            return np.full((rows, 1), float(v))
        return v

    def constrain_initial_state(self, initial_state):
        self.subject_to(self.initial_state() == self._column(initial_state, self._X.rows()))

    def constrain_final_state(self, final_state):
        self.subject_to(self.final_state() == self._column(final_state, self._X.rows()))

    def for_each_step(self, callback):
        """ocp.hpp:183-213: callback(x, u) sees every sample, the last one included"""
        for k in range(self._samples):
            callback(self._X[:, k:k + 1], self._U[:, k:k + 1])

    def set_lower_input_bound(self, lower_bound):
        for k in range(self._samples):
            self.subject_to(self._U[:, k:k + 1] >= self._column(lower_bound, self._U.rows()))

    def set_upper_input_bound(self, upper_bound):
        for k in range(self._samples):
            self.subject_to(self._U[:, k:k + 1] <= self._column(upper_bound, self._U.rows()))

    def set_min_timestep(self, min_timestep):
        self.subject_to(self._DT >= float(min_timestep))

    def set_max_timestep(self, max_timestep):
        self.subject_to(self._DT <= float(max_timestep))

    def X(self): return self._X
    def U(self): return self._U
    def dt(self): return self._DT
    def initial_state(self): return self._X[:, 0:1]
    def final_state(self): return self._X[:, self._samples - 1:self._samples]


def multistart(solve, initial_guesses):
    """python/src/sleipnir/optimization/__init__.py:6-31 (multistart.hpp:45-79): every initial
    guess on its own thread — the expression graph of libslpx is per thread, so `solve` builds its
    problem inside the call like the reference's tests do — then successful solves first, lowest
    cost among them.  `solve(guess)` returns (status, cost, variables)."""
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(initial_guesses)) as executor:
        futures = [executor.submit(solve, guess) for guess in initial_guesses]
        results = [f.result() for f in concurrent.futures.as_completed(futures)]
    return min(results, key=lambda r: (int(r[0] != ExitStatus.SUCCESS), r[1]))
