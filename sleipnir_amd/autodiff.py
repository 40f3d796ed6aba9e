"""The reference's Python modelling interface `sleipnir.autodiff`
(python/cpp/autodiff/bind_variable.cpp:92-178, bind_variable_matrix.cpp, bind_gradient.cpp,
bind_jacobian.cpp, bind_hessian.cpp, bind_expression_type.cpp) over the C-ABI of libslpx:
same names, argument meaning and behaviour, so that a program written against the reference's
binding runs on the MI355X path by changing its import (sleipnir_amd.compat.install() even
registers the reference's module names).

The reference binds its C++ classes with nanobind; here the expression graph lives in libslpx
(slpx_expr_*, include/slpx.h:60-71) and this module is the host-side mirror above the C-ABI:
a Variable is a node id, a VariableMatrix a 2-D numpy object array of Variables (so a slice IS
a view and assigning into it writes through, like VariableBlock).  Gradient / Jacobian /
Hessian evaluate through the compiled tape on the GPU (no CPU fallback: they raise without a
HIP device); their get() is the symbolic gradient tree like the reference's.
"""
from __future__ import annotations

import enum
import math as _math
import numbers

import numpy as np

import sleipnir_amd as _sa

__all__ = [
    "ExpressionType", "Variable", "VariableMatrix", "VariableBlock", "Gradient", "Jacobian", "Hessian",
    "abs", "acos", "asin", "atan", "atan2", "cbrt", "cos", "cosh", "erf", "exp", "hypot", "log", "log10",
    "max", "min", "pow", "sign", "sin", "sinh", "sqrt", "tan", "tanh", "cwise_reduce", "block", "solve",
]

_OPS = _sa.OPS
_builtin_abs, _builtin_max, _builtin_min, _builtin_pow = abs, max, min, pow


class ExpressionType(enum.IntEnum):
    """expression_type.hpp:15-26"""
    NONE = 0
    CONSTANT = 1
    LINEAR = 2
    QUADRATIC = 3
    NONLINEAR = 4


class _Lib:
    """Entry points looked up once: a model is built one ctypes call per node."""
    _fns = None

    @classmethod
    def get(cls):
        if cls._fns is None:
            L = _sa.lib()
            cls._fns = (L.slpx_expr_variable, L.slpx_expr_constant, L.slpx_expr_unary, L.slpx_expr_binary,
                        L.slpx_expr_type, L.slpx_expr_value, L.slpx_expr_set_value)
        return cls._fns


def _is_scalar(v):
    return isinstance(v, numbers.Real) and not isinstance(v, bool) or isinstance(v, np.generic)


class Variable:
    """variable.hpp:52-300: a handle on an expression node.  Variable() is a free variable of
    value 0, Variable(number) a constant."""
    __slots__ = ("node",)
    __array_ufunc__ = None  # numpy defers to the reflected operators below

    def __init__(self, value=None):
        if value is None:
            self.node = _Lib.get()[0](0.0)
        elif isinstance(value, Variable):
            self.node = value.node
        elif isinstance(value, VariableMatrix):
            if value.shape != (1, 1):
                raise ValueError("Variable(VariableMatrix) needs a 1x1 matrix")
            self.node = value[0, 0].node
        else:
            self.node = _Lib.get()[1](float(value))

    @staticmethod
    def _wrap(node: int) -> "Variable":
        v = Variable.__new__(Variable)
        v.node = node
        return v

    @staticmethod
    def _lift(v) -> "Variable":
        if isinstance(v, Variable):
            return v
        if isinstance(v, VariableMatrix):
            return Variable(v)
        if isinstance(v, np.ndarray):
            if v.size != 1:
                raise ValueError("expected a scalar")
            v = v.reshape(())[()]
        return Variable._wrap(_Lib.get()[1](float(v)))

    def set_value(self, value):
        _Lib.get()[6](self.node, float(value))

    def value(self) -> float:
        return _Lib.get()[5](self.node)

    def type(self) -> ExpressionType:
        return ExpressionType(_Lib.get()[4](self.node))

    def _bin(self, op, other, swap=False):
        if isinstance(other, (VariableMatrix, np.ndarray)) and not (isinstance(other, np.ndarray) and other.size == 1):
            return NotImplemented
        o = Variable._lift(other)
        a, b = (o.node, self.node) if swap else (self.node, o.node)
        return Variable._wrap(_Lib.get()[3](_OPS[op], a, b))

    def __add__(self, o): return self._bin("ADD", o)
    def __radd__(self, o): return self._bin("ADD", o, True)
    def __sub__(self, o): return self._bin("SUB", o)
    def __rsub__(self, o): return self._bin("SUB", o, True)
    def __mul__(self, o): return self._bin("MUL", o)
    def __rmul__(self, o): return self._bin("MUL", o, True)
    def __truediv__(self, o): return self._bin("DIV", o)
    def __rtruediv__(self, o): return self._bin("DIV", o, True)
    def __pow__(self, o): return self._bin("POW", o)
    def __rpow__(self, o): return self._bin("POW", o, True)
    def __neg__(self): return Variable._wrap(_Lib.get()[2](_OPS["NEG"], self.node))
    def __pos__(self): return self

    # variable.hpp:716-778: a comparison is a constraint (whose truth value is its value now)
    def __eq__(self, o): return _constraints("eq", self, o)  # type: ignore[override]
    def __le__(self, o): return _constraints("ge", o, self)
    def __lt__(self, o): return _constraints("ge", o, self)
    def __ge__(self, o): return _constraints("ge", self, o)
    def __gt__(self, o): return _constraints("ge", self, o)
    __hash__ = object.__hash__

    def __repr__(self):
        return f"Variable({self.value()!r})"


def _unary(op):
    def fn(x):
        if isinstance(x, VariableMatrix):
            return x.cwise_map(fn)
        return Variable._wrap(_Lib.get()[2](_OPS[op], Variable._lift(x).node))
    fn.__name__ = op.lower()
    fn.__doc__ = f"slp::{op.lower()} (variable.hpp)"
    return fn


def _binary(op):
    def fn(a, b):
        return Variable._wrap(_Lib.get()[3](_OPS[op], Variable._lift(a).node, Variable._lift(b).node))
    fn.__name__ = op.lower()
    fn.__doc__ = f"slp::{op.lower()} (variable.hpp)"
    return fn


abs, acos, asin, atan, cbrt = _unary("ABS"), _unary("ACOS"), _unary("ASIN"), _unary("ATAN"), _unary("CBRT")
cos, cosh, erf, exp, log = _unary("COS"), _unary("COSH"), _unary("ERF"), _unary("EXP"), _unary("LOG")
log10, sign, sin, sinh, sqrt = _unary("LOG10"), _unary("SIGN"), _unary("SIN"), _unary("SINH"), _unary("SQRT")
tan, tanh = _unary("TAN"), _unary("TANH")
atan2, max, min, pow = _binary("ATAN2"), _binary("MAX"), _binary("MIN"), _binary("POW")
_hypot2 = _binary("HYPOT")


def hypot(x, y, z=None):
    """variable.hpp:695-714"""
    if z is None:
        return _hypot2(x, y)
    return sqrt(pow(x, 2) + pow(y, 2) + pow(z, 2))


class VariableMatrix:
    """variable_matrix.hpp:43-1160 (and variable_block.hpp: a slice is a VariableMatrix that
    shares its parent's storage)."""
    __slots__ = ("_a",)
    __array_ufunc__ = None

    def __init__(self, *args):
        if len(args) == 0:
            self._a = np.empty((0, 0), dtype=object)
        elif len(args) == 2 or (len(args) == 1 and isinstance(args[0], numbers.Integral)
                                and not isinstance(args[0], bool)):
            rows, cols = (int(args[0]), int(args[1])) if len(args) == 2 else (int(args[0]), 1)
            self._a = np.empty((rows, cols), dtype=object)
            for r in range(rows):
                for c in range(cols):
                    self._a[r, c] = Variable()
        elif len(args) == 1:
            v = args[0]
            if isinstance(v, VariableMatrix):
                self._a = v._a.copy()
            elif isinstance(v, Variable):
                self._a = np.empty((1, 1), dtype=object)
                self._a[0, 0] = v
            else:
                rows = [list(row) if isinstance(row, (list, tuple, np.ndarray)) else [row] for row in
                        (v.tolist() if isinstance(v, np.ndarray) else v)]
                ncol = len(rows[0]) if rows else 0
                self._a = np.empty((len(rows), ncol), dtype=object)
                for r, row in enumerate(rows):
                    if len(row) != ncol:
                        raise ValueError("rows of different lengths")
                    for c, e in enumerate(row):
                        self._a[r, c] = Variable._lift(e)
        else:
            raise TypeError("VariableMatrix(), (rows), (rows, cols) or (list of lists)")

    @staticmethod
    def _of(a: np.ndarray) -> "VariableMatrix":
        m = VariableMatrix.__new__(VariableMatrix)
        m._a = a
        return m

    @staticmethod
    def _lift(v) -> "VariableMatrix":
        if isinstance(v, VariableMatrix):
            return v
        if isinstance(v, Variable) or _is_scalar(v):
            return VariableMatrix(Variable._lift(v))
        a = np.asarray(v)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return VariableMatrix(a)

    # ---- shape ----
    def rows(self): return self._a.shape[0]
    def cols(self): return self._a.shape[1]
    @property
    def shape(self): return self._a.shape
    def __len__(self): return self._a.shape[0]
    def __iter__(self): return iter(self._a.reshape(-1).tolist())

    # ---- element / slice access (bind_variable_matrix.cpp: __getitem__, __setitem__) ----
    @staticmethod
    def _index(key, rows, cols):
        if not isinstance(key, tuple):
            i = int(key)
            if i < 0:
                i += rows * cols
            if not 0 <= i < rows * cols:
                raise IndexError("index out of bounds")
            return divmod(i, cols) if cols else (0, 0)
        if len(key) != 2:
            raise IndexError(f"Expected 2 slices, got {len(key)}.")
        return key

    def __getitem__(self, key):
        r, c = self._index(key, *self._a.shape)
        r_int, c_int = not isinstance(r, slice), not isinstance(c, slice)
        if r_int and c_int:
            return self._a[int(r), int(c)]
        # an integer beside a slice keeps its dimension (a column stays a column)
        rs = slice(int(r), int(r) + 1 or None) if r_int else r
        cs = slice(int(c), int(c) + 1 or None) if c_int else c
        return VariableMatrix._of(self._a[rs, cs])

    def __setitem__(self, key, value):
        r, c = self._index(key, *self._a.shape)
        if not isinstance(r, slice) and not isinstance(c, slice):
            self._a[int(r), int(c)] = Variable._lift(value)
            return
        target = self[r, c]._a
        if isinstance(value, VariableMatrix) and value._a is target:
            return  # `M[a:b, c:d] += X` assigns the view to itself
        v = VariableMatrix._lift(value)
        if v.shape != target.shape:
            if v.shape == (1, 1):
                v = VariableMatrix._of(np.full(target.shape, v._a[0, 0], dtype=object))
            else:
                raise ValueError(f"shape mismatch: {v.shape} into {target.shape}")
        target[...] = v._a

    def row(self, r): return self[r:r + 1 or None, :]
    def col(self, c): return self[:, c:c + 1 or None]
    def block(self, row_offset, col_offset, block_rows, block_cols):
        return VariableMatrix._of(self._a[row_offset:row_offset + block_rows, col_offset:col_offset + block_cols])

    def segment(self, offset, length):
        if self._a.shape[0] == 1 and self._a.shape[1] != 1:
            return self.block(0, offset, 1, length)
        return self.block(offset, 0, length, 1)

    @property
    def T(self): return VariableMatrix._of(self._a.T.copy())  # (a new matrix, variable_matrix.hpp:955-965: not a view)

    # ---- values ----
    def value(self, *idx):
        if len(idx) == 0:
            out = np.empty(self._a.shape)
            for r in range(out.shape[0]):
                for c in range(out.shape[1]):
                    out[r, c] = self._a[r, c].value()
            return out
        return self[idx[0] if len(idx) == 1 else idx].value()

    def set_value(self, values):
        v = np.asarray(values, dtype=np.float64)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        if v.shape != self._a.shape:
            raise ValueError(f"shape mismatch: {v.shape} into {self._a.shape}")
        for r in range(v.shape[0]):
            for c in range(v.shape[1]):
                self._a[r, c].set_value(v[r, c])

    def cwise_map(self, unary_op):
        """variable_matrix.hpp:1027-1039 (cwise_transform; the binding's name is cwise_map)"""
        out = np.empty(self._a.shape, dtype=object)
        for r in range(out.shape[0]):
            for c in range(out.shape[1]):
                out[r, c] = Variable._lift(unary_op(self._a[r, c]))
        return VariableMatrix._of(out)

    cwise_transform = cwise_map

    def exp(self):
        """variable_matrix.hpp:1044-1098: the matrix exponential, q(A)^-1 p(A) with p the degree-13
        diagonal Pade numerator and q(A) = p(-A); c_k = c_{k-1} (m - k + 1) / (k (2m - k + 1))."""
        n = self.rows()
        if n != self.cols():
            raise ValueError("exp() needs a square matrix")
        m = 13
        c = [1.0]
        for k in range(1, m + 1):
            c.append(c[-1] * (m - k + 1) / (k * (2 * m - k + 1)))
        I = np.eye(n)
        P, Q = VariableMatrix(I * c[m]), VariableMatrix(I * -c[m])
        for k in range(m - 1, -1, -1):
            P = P @ self + I * c[k]
            Q = Q @ self + I * (-c[k] if k & 1 else c[k])
        return solve(Q, P)

    # ---- arithmetic (variable_matrix.hpp:587-1020) ----
    @staticmethod
    def _matmul(a: np.ndarray, b: np.ndarray) -> "VariableMatrix":
        if a.shape[1] != b.shape[0]:
            raise ValueError(f"matrix product of {a.shape} and {b.shape}")
        out = np.empty((a.shape[0], b.shape[1]), dtype=object)
        for i in range(a.shape[0]):
            for j in range(b.shape[1]):
                acc = a[i, 0] * b[0, j]
                for k in range(1, a.shape[1]):
                    acc = acc + a[i, k] * b[k, j]
                out[i, j] = acc
        return VariableMatrix._of(out)

    def _cwise(self, other, f):
        o = VariableMatrix._lift(other)
        if o.shape != self.shape:
            if o.shape == (1, 1):  # matrix (+|-) scalar
                o = VariableMatrix._of(np.full(self.shape, o._a[0, 0], dtype=object))
            elif self.shape == (1, 1):
                return o._cwise(self, lambda x, y: f(y, x))
            else:
                raise ValueError(f"shape mismatch: {self.shape} and {o.shape}")
        out = np.empty(self.shape, dtype=object)
        for r in range(out.shape[0]):
            for c in range(out.shape[1]):
                out[r, c] = f(self._a[r, c], o._a[r, c])
        return VariableMatrix._of(out)

    def __add__(self, o): return self._cwise(o, lambda x, y: x + y)
    def __radd__(self, o): return self._cwise(o, lambda x, y: y + x)
    def __sub__(self, o): return self._cwise(o, lambda x, y: x - y)
    def __rsub__(self, o): return self._cwise(o, lambda x, y: y - x)
    def __neg__(self): return self.cwise_map(lambda x: -x)
    def __pos__(self): return self

    def __mul__(self, o):
        if isinstance(o, Variable) or _is_scalar(o):
            s = Variable._lift(o)
            return self.cwise_map(lambda x: x * s)
        return VariableMatrix._matmul(self._a, VariableMatrix._lift(o)._a)

    def __rmul__(self, o):
        if isinstance(o, Variable) or _is_scalar(o):
            s = Variable._lift(o)
            return self.cwise_map(lambda x: s * x)
        return VariableMatrix._matmul(VariableMatrix._lift(o)._a, self._a)

    def __matmul__(self, o): return VariableMatrix._matmul(self._a, VariableMatrix._lift(o)._a)
    def __rmatmul__(self, o): return VariableMatrix._matmul(VariableMatrix._lift(o)._a, self._a)

    def __truediv__(self, o):
        s = Variable._lift(o)
        return self.cwise_map(lambda x: x / s)

    def __pow__(self, p):
        if self.shape != (1, 1):
            raise ValueError("** needs a 1x1 matrix")
        return self._a[0, 0] ** p

    # in place: the handles are re-pointed, so views of this storage see the result
    def _assign(self, result):
        self._a[...] = result._a
        return self

    def __iadd__(self, o): return self._assign(self + o)
    def __isub__(self, o): return self._assign(self - o)
    def __imul__(self, o): return self._assign(self * o)
    def __itruediv__(self, o): return self._assign(self / o)

    # ---- constraints ----
    def __eq__(self, o): return _constraints("eq", self, o)  # type: ignore[override]
    def __le__(self, o): return _constraints("ge", o, self)
    def __lt__(self, o): return _constraints("ge", o, self)
    def __ge__(self, o): return _constraints("ge", self, o)
    def __gt__(self, o): return _constraints("ge", self, o)
    __hash__ = object.__hash__

    # ---- static constructors (variable_matrix.hpp:1100-1160) ----
    @staticmethod
    def constant(rows, cols, value):
        return VariableMatrix(np.full((rows, cols), float(value)))

    @staticmethod
    def zero(rows, cols): return VariableMatrix.constant(rows, cols, 0.0)
    @staticmethod
    def one(rows, cols): return VariableMatrix.constant(rows, cols, 1.0)
    @staticmethod
    def identity(rows, cols=None): return VariableMatrix(np.eye(rows, rows if cols is None else cols))

    def __repr__(self):
        return f"VariableMatrix({self.value()!r})"


VariableBlock = VariableMatrix  # a slice shares storage with its matrix: one class serves both


def cwise_reduce(lhs: VariableMatrix, rhs: VariableMatrix, binary_op) -> VariableMatrix:
    """variable_matrix.hpp:1378-1395"""
    return VariableMatrix._lift(lhs)._cwise(rhs, lambda x, y: Variable._lift(binary_op(x, y)))


def block(list_of_rows) -> VariableMatrix:
    """variable_matrix.hpp:1397-1470: [[A, B], [C]] -> one matrix"""
    rows = []
    for blocks in list_of_rows:
        mats = [VariableMatrix._lift(b)._a for b in blocks]
        if len({m.shape[0] for m in mats}) != 1:
            raise ValueError("blocks of one row must have the same height")
        rows.append(np.concatenate(mats, axis=1))
    if len({r.shape[1] for r in rows}) != 1:
        raise ValueError("block rows must have the same width")
    return VariableMatrix._of(np.concatenate(rows, axis=0))


def solve(A: VariableMatrix, B: VariableMatrix) -> VariableMatrix:
    """variable_matrix.hpp:1480-1580: X with A X = B — closed forms up to 3x3 like the reference,
    Gaussian elimination on expressions beyond (the reference goes through Eigen there)."""
    A, B = VariableMatrix._lift(A), VariableMatrix._lift(B)
    n = A.rows()
    if A.cols() != n or B.rows() != n:
        raise ValueError("solve(A, B): A must be square with as many rows as B")
    a = A._a
    if n == 1:
        return B / a[0, 0]
    if n == 2:
        det = a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
        adj = VariableMatrix([[a[1, 1], -a[0, 1]], [-a[1, 0], a[0, 0]]])
        return (adj @ B) / det
    if n == 3:
        (p, q, r), (s, t, u), (v, w, x) = a.tolist()
        c00, c10, c20 = t * x - u * w, u * v - s * x, s * w - t * v
        det = p * c00 + q * c10 + r * c20
        adj = VariableMatrix([[c00, r * w - q * x, q * u - r * t], [c10, p * x - r * v, r * s - p * u],
                              [c20, q * v - p * w, p * t - q * s]])
        return (adj @ B) / det
    m = np.concatenate([a.copy(), B._a.copy()], axis=1)
    for k in range(n):
        for i in range(k + 1, n):
            f = m[i, k] / m[k, k]
            for j in range(k, m.shape[1]):
                m[i, j] = m[i, j] - f * m[k, j]
    out = np.empty(B.shape, dtype=object)
    for j in range(B.cols()):
        for i in reversed(range(n)):
            acc = m[i, n + j]
            for k in range(i + 1, n):
                acc = acc - m[i, k] * out[k, j]
            out[i, j] = acc / m[i, i]
    return VariableMatrix._of(out)


# ---------------------------------------------------------------------------------------------
# constraints (variable.hpp:716-1013); sleipnir_amd.optimization re-exports them
# ---------------------------------------------------------------------------------------------
class EqualityConstraints:
    """variable.hpp:832-870: a vector of expressions each constrained to be 0"""

    def __init__(self, constraints=()):
        self.constraints = []
        for c in constraints:
            self.constraints.extend(c.constraints if isinstance(c, EqualityConstraints) else [c])

    def __bool__(self):
        return all(c.value() == 0.0 for c in self.constraints)


class InequalityConstraints:
    """variable.hpp:877-915: a vector of expressions each constrained to be >= 0"""

    def __init__(self, constraints=()):
        self.constraints = []
        for c in constraints:
            self.constraints.extend(c.constraints if isinstance(c, InequalityConstraints) else [c])

    def __bool__(self):
        return all(c.value() >= 0.0 for c in self.constraints)


def _constraints(kind, lhs, rhs):
    """lhs - rhs, element by element; a scalar beside a matrix applies to every element
    (variable.hpp:716-778)."""
    if isinstance(lhs, VariableMatrix) or isinstance(rhs, VariableMatrix) or isinstance(lhs, np.ndarray) or isinstance(
            rhs, np.ndarray):
        l, r = VariableMatrix._lift(lhs), VariableMatrix._lift(rhs)
        if l.shape != r.shape and (1, 1) not in (l.shape, r.shape):
            raise ValueError(f"shape mismatch: {l.shape} and {r.shape}")
        diff = (l - r)._a.reshape(-1).tolist()
    else:
        diff = [Variable._lift(lhs) - Variable._lift(rhs)]
    out = EqualityConstraints() if kind == "eq" else InequalityConstraints()
    out.constraints = diff
    return out


# ---------------------------------------------------------------------------------------------
# Gradient, Jacobian, Hessian (gradient.hpp:24-77, jacobian.hpp:30-170, hessian.hpp:33-170)
# ---------------------------------------------------------------------------------------------
def _nodes(wrt):
    if isinstance(wrt, Variable):
        return [wrt.node]
    return [v.node for v in VariableMatrix._lift(wrt)]


def _gradient_tree(f_node, wrt_nodes):
    w = np.asarray(wrt_nodes, dtype=np.int32)
    out = np.zeros(len(w), dtype=np.int32)
    _sa.lib().slpx_expr_gradient_tree(int(f_node), w.ctypes.data, len(w), out.ctypes.data)
    return [Variable._wrap(int(n)) if n >= 0 else Variable(0.0) for n in out]  # < 0: structurally zero


class _Evaluator:
    """cost and/or equality rows over `wrt` as the decision variables of a throw-away problem:
    NLP structure and tape are compiled once, value() re-runs the sweep kernels on the GPU at the
    variables' current values and reads g / A_e / H out of the value vector."""

    def __init__(self, cost, rows, wrt):
        import ctypes

        import scipy.sparse  # noqa: F401  (value() returns scipy matrices like the reference)

        L = _sa.lib()
        self._wrt = list(wrt)
        self._n_rows = len(rows)
        self._p = _sa.Problem()
        if not hasattr(L, "_adopt_ready"):
            L.slpx_problem_adopt_variable.restype = None
            L.slpx_problem_adopt_variable.argtypes = [ctypes.c_void_p, ctypes.c_int32]
            L._adopt_ready = True
        for w in self._wrt:
            L.slpx_problem_adopt_variable(self._p._h, int(w))
        if cost is not None:
            self._p.minimize(int(cost))
        for r in rows:
            self._p.subject_to_eq(int(r))
        self._sys = None

    def sweep(self):
        if self._sys is None:
            self._sys = _sa.System(self._p, batch=1, device=0)  # raises without a HIP device
        get = _Lib.get()[5]
        x = np.array([get(int(w)) for w in self._wrt])
        s = self._sys
        s.set_state(x, np.ones(1), np.zeros(_builtin_max(1, self._n_rows)), np.ones(1), np.array([0.1]))
        s.sweep(True)
        return s.info, s.get("V")[0]

    def close(self):
        if self._sys is not None:
            self._sys.close()
            self._sys = None
        if self._p is not None:
            self._p.close()
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gradient:
    """gradient.hpp:24-77"""

    def __init__(self, variable, wrt):
        self._f = Variable._lift(variable)
        self._wrt = _nodes(wrt)
        self._ev = None

    def get(self) -> VariableMatrix:
        return VariableMatrix([[g] for g in _gradient_tree(self._f.node, self._wrt)])

    def value(self):
        import scipy.sparse

        if self._ev is None:
            self._ev = _Evaluator(self._f.node, [], self._wrt)
        info, V = self._ev.sweep()
        cp, ri = self._ev._sys.pattern(0)
        n = len(self._wrt)
        out = np.zeros(n)
        for c in range(n):
            for q in range(cp[c], cp[c + 1]):
                out[c] += V[info["off_g"] + q]
        return scipy.sparse.csc_matrix(out.reshape(-1, 1))


class Jacobian:
    """jacobian.hpp:30-170"""

    def __init__(self, variables, wrt):
        self._rows = _nodes(variables)
        self._wrt = _nodes(wrt)
        self._ev = None

    def get(self) -> VariableMatrix:
        return VariableMatrix([_gradient_tree(r, self._wrt) for r in self._rows])

    def value(self):
        import scipy.sparse

        if self._ev is None:
            self._ev = _Evaluator(None, self._rows, self._wrt)
        info, V = self._ev.sweep()
        cp, ri = self._ev._sys.pattern(1)
        nnz = int(cp[-1])
        data = np.asarray(V[info["off_Ae"]:info["off_Ae"] + nnz], dtype=np.float64)
        m = scipy.sparse.csc_matrix((data, np.asarray(ri[:nnz]), np.asarray(cp)), shape=(len(self._rows), len(self._wrt)))
        m.sum_duplicates()
        return m


class Hessian:
    """hessian.hpp:33-170: both triangles, like the reference's default"""

    def __init__(self, variable, wrt):
        self._f = Variable._lift(variable)
        self._wrt = _nodes(wrt)
        self._ev = None

    def get(self) -> VariableMatrix:
        g = _gradient_tree(self._f.node, self._wrt)
        return VariableMatrix([_gradient_tree(gi.node, self._wrt) for gi in g])

    def value(self):
        import scipy.sparse

        if self._ev is None:
            self._ev = _Evaluator(self._f.node, [], self._wrt)
        info, V = self._ev.sweep()
        cp, ri = self._ev._sys.pattern(3)
        n = len(self._wrt)
        nnz = int(cp[-1])
        data = np.asarray(V[info["off_Hf"]:info["off_Hf"] + nnz], dtype=np.float64)
        lower = scipy.sparse.csc_matrix((data, np.asarray(ri[:nnz]), np.asarray(cp)), shape=(n, n))
        lower.sum_duplicates()
        strict = scipy.sparse.tril(lower, k=-1)
        return (lower + strict.T).tocsc()
