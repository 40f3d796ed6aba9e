"""ctypes wrapper over oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference algorithm (see oracle/*.hpp
headers).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
ORACLE_DIR = ROOT / "oracle"
LIB_PATH = ORACLE_DIR / "liboracle.so"

# orc::Op (oracle/ad.hpp) — same numbering as slpx_op
OPS = {name: i for i, name in enumerate([
    "CONST", "VAR", "ADD", "SUB", "NEG", "MUL", "DIV", "POW", "ABS", "SIGN", "SQRT", "CBRT", "EXP",
    "LOG", "LOG10", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "ATAN2", "SINH", "COSH", "TANH",
    "ERF", "HYPOT", "MAX", "MIN", "ISNONNEG", "ISPOS"])}


def build():
    res = subprocess.run(["make", "-C", str(ORACLE_DIR)], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building liboracle.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = ctypes.CDLL(str(LIB_PATH))
    vp, i, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double

    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("orc_reset", None)
    sig("orc_var", i, d)
    sig("orc_const", i, d)
    sig("orc_unary", i, i, i)
    sig("orc_binary", i, i, i, i)
    sig("orc_value", d, i)
    sig("orc_set_value", None, i, d)
    sig("orc_type", i, i)
    sig("orc_opcode", i, i)
    sig("orc_num_nodes", ctypes.c_long)
    sig("orc_jacobian", None, vp, i, vp, i, vp)
    sig("orc_gradient", None, i, vp, i, vp)
    sig("orc_hessian", None, i, vp, i, i, vp)
    sig("orc_gradient_tree", None, i, vp, i, vp)
    sig("orc_problem_new", i)
    sig("orc_problem_decision_variable", i, i)
    sig("orc_problem_minimize", None, i, i)
    sig("orc_problem_maximize", None, i, i)
    sig("orc_problem_subject_to_eq", None, i, i)
    sig("orc_problem_subject_to_ineq", None, i, i)
    sig("orc_problem_cost_type", i, i)
    sig("orc_problem_eq_type", i, i)
    sig("orc_problem_ineq_type", i, i)
    sig("orc_problem_dims", None, i, vp, vp, vp)
    sig("orc_problem_get_x", None, i, vp)
    sig("orc_problem_set_x", None, i, vp)
    sig("orc_problem_get_duals", None, i, vp, vp, vp)
    sig("orc_problem_solve", i, i, d, i, d, vp, i, vp)
    sig("orc_problem_solve_trace", i, i, d, i, vp, i, i, vp, vp)
    sig("orc_problem_restoration_steps", i, i, d, i, vp, vp, vp, vp, d, i)
    sig("orc_build_cart_pole", i, i, d)
    sig("orc_build_flywheel", i, i, d)
    sig("orc_problem_scaling", None, i, vp, vp, vp)
    sig("orc_problem_newton_step", i, i, vp, vp, vp, vp, d, i, vp, i, i, vp)
    sig("orc_eval_f", d, i)
    sig("orc_eval_reg", None, i, vp, vp, vp, vp)
    sig("orc_eval_csc_nnz", i, i, i)
    sig("orc_eval_csc", None, i, i, vp, vp, vp)
    sig("orc_eval_vec_len", i, i, i)
    sig("orc_eval_vec", None, i, i, vp)
    sig("orc_eval_perm", None, i, vp)
    sig("orc_ldlt_solve", i, i, i, i, vp, vp, vp, vp, vp, i, d, d, vp, vp, vp, vp)
    sig("orc_ldlt_factor_once", i, i, i, i, vp, vp, vp, vp, i, d, d, vp, vp)
    _lib = L
    return L


def _ia(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fa(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleProblem:
    """A problem inside the oracle (oracle/problem.hpp)."""

    CSC = {"A_e": 0, "A_i": 1, "H": 2, "lhs": 3}
    VEC = {"g": 0, "c_e": 1, "c_i": 2, "rhs": 3, "p_x": 4, "p_y": 5, "p_s": 6, "p_z": 7, "D": 8, "p": 9}

    def __init__(self, pid: int):
        self.pid = pid

    @classmethod
    def new(cls):
        return cls(lib().orc_problem_new())

    @classmethod
    def cart_pole(cls, N, dt):
        return cls(lib().orc_build_cart_pole(N, dt))

    @classmethod
    def flywheel(cls, N, dt):
        return cls(lib().orc_build_flywheel(N, dt))

    @property
    def dims(self):
        n, me, mi = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().orc_problem_dims(self.pid, ctypes.addressof(n), ctypes.addressof(me), ctypes.addressof(mi))
        return n.value, me.value, mi.value

    def types(self):
        L = lib()
        return L.orc_problem_cost_type(self.pid), L.orc_problem_eq_type(self.pid), L.orc_problem_ineq_type(self.pid)

    def get_x(self):
        x = np.zeros(self.dims[0])
        lib().orc_problem_get_x(self.pid, x.ctypes.data)
        return x

    def set_x(self, x):
        x = _fa(x)
        lib().orc_problem_set_x(self.pid, x.ctypes.data)

    def duals(self):
        n, me, mi = self.dims
        s, y, z = np.zeros(mi), np.zeros(me), np.zeros(mi)
        lib().orc_problem_get_duals(self.pid, s.ctypes.data, y.ctypes.data, z.ctypes.data)
        return s, y, z

    def scaling(self):
        n, me, mi = self.dims
        df = ctypes.c_double()
        ce, ci = np.zeros(max(me, 1)), np.zeros(max(mi, 1))
        lib().orc_problem_scaling(self.pid, ctypes.addressof(df), ce.ctypes.data, ci.ctypes.data)
        return np.concatenate([[df.value], ce[:me], ci[:mi]])

    def solve(self, tolerance=1e-8, max_iterations=5000, timeout=0.0, perm=None):
        stats = np.zeros(9)
        p = None if perm is None else _ia(perm)
        status = lib().orc_problem_solve(self.pid, tolerance, max_iterations, timeout,
                                         None if p is None else p.ctypes.data,
                                         0 if p is None else len(p), stats.ctypes.data)
        keys = ["iterations", "factorizations", "solves", "t_ad", "t_build", "t_decomp", "t_solve",
                "t_linesearch", "t_total"]
        return status, dict(zip(keys, stats))

    def solve_trace(self, tolerance=1e-8, max_iterations=5000, perm=None, max_records=6000):
        """solve() recording, per iteration, [iteration, len(x), |x|_2, |s|_2, |y|_2, |z|_2] (x, s:
        the outer problem's part of the iterate, also inside feasibility restoration)."""
        out = np.zeros((max_records, 6))
        nrec = ctypes.c_int()
        p = None if perm is None else _ia(perm)
        status = lib().orc_problem_solve_trace(self.pid, tolerance, max_iterations,
                                               None if p is None else p.ctypes.data, 0 if p is None else len(p),
                                               max_records, out.ctypes.data, ctypes.addressof(nrec))
        return status, out[:nrec.value]

    def restoration_steps(self, x, s, y, z, mu, steps, tolerance=1e-8, max_iterations=5000):
        """feasibility_restoration from the given iterate, `steps` iterations (oracle/ipm.hpp)."""
        x, s, y, z = (np.array(a, dtype=np.float64, copy=True) for a in (x, s, y, z))
        status = lib().orc_problem_restoration_steps(self.pid, tolerance, max_iterations, x.ctypes.data,
                                                     s.ctypes.data, y.ctypes.data, z.ctypes.data, float(mu), int(steps))
        return status, x, s, y, z

    def newton_step(self, x, s, y, z, mu, do_solve=True, perm=None, reuse_solver=False):
        x, s, y, z = _fa(x), _fa(s), _fa(y), _fa(z)
        p = None if perm is None else _ia(perm)
        timing = np.zeros(4)
        info = lib().orc_problem_newton_step(self.pid, x.ctypes.data, s.ctypes.data, y.ctypes.data,
                                             z.ctypes.data, mu, int(do_solve),
                                             None if p is None else p.ctypes.data,
                                             0 if p is None else len(p), int(reuse_solver),
                                             timing.ctypes.data)
        return info, dict(zip(["t_ad", "t_build", "t_decomp", "t_solve"], timing))

    def f(self):
        return lib().orc_eval_f(self.pid)

    def reg(self):
        d, g = ctypes.c_double(), ctypes.c_double()
        nf, nl = ctypes.c_int(), ctypes.c_int()
        lib().orc_eval_reg(self.pid, ctypes.addressof(d), ctypes.addressof(g), ctypes.addressof(nf),
                           ctypes.addressof(nl))
        return d.value, g.value, nf.value, nl.value

    def csc(self, name):
        which = self.CSC[name]
        n, me, mi = self.dims
        ncols = n + me if name == "lhs" else n
        nnz = lib().orc_eval_csc_nnz(self.pid, which)
        colptr = np.zeros(ncols + 1, dtype=np.int32)
        rowidx = np.zeros(max(nnz, 1), dtype=np.int32)
        val = np.zeros(max(nnz, 1))
        lib().orc_eval_csc(self.pid, which, colptr.ctypes.data, rowidx.ctypes.data, val.ctypes.data)
        return colptr, rowidx[:nnz], val[:nnz]

    def vec(self, name):
        which = self.VEC[name]
        ln = lib().orc_eval_vec_len(self.pid, which)
        out = np.zeros(max(ln, 1))
        lib().orc_eval_vec(self.pid, which, out.ctypes.data)
        return out[:ln]

    def perm(self):
        n, me, mi = self.dims
        p = np.zeros(n + me, dtype=np.int32)
        lib().orc_eval_perm(self.pid, p.ctypes.data)
        return p


def ldlt_factor_once(n, m_e, colptr, rowidx, val, delta, gamma, perm=None):
    colptr, rowidx, val = _ia(colptr), _ia(rowidx), _fa(val)
    nt = n + m_e
    D = np.zeros(nt)
    inertia = np.zeros(3, dtype=np.int32)
    p = None if perm is None else _ia(perm)
    info = lib().orc_ldlt_factor_once(nt, n, m_e, colptr.ctypes.data, rowidx.ctypes.data,
                                      val.ctypes.data, None if p is None else p.ctypes.data,
                                      0 if p is None else len(p), delta, gamma, D.ctypes.data,
                                      inertia.ctypes.data)
    return info, D, inertia


def ldlt_solve(n, m_e, colptr, rowidx, val, rhs, perm=None, gamma_min=1e-10):
    colptr, rowidx, val, rhs = _ia(colptr), _ia(rowidx), _fa(val), _fa(rhs)
    nt = n + m_e
    x, D, dg = np.zeros(nt), np.zeros(nt), np.zeros(2)
    nf = ctypes.c_int()
    p = None if perm is None else _ia(perm)
    info = lib().orc_ldlt_solve(nt, n, m_e, colptr.ctypes.data, rowidx.ctypes.data, val.ctypes.data,
                                rhs.ctypes.data, None if p is None else p.ctypes.data,
                                0 if p is None else len(p), gamma_min, 0.0, x.ctypes.data,
                                D.ctypes.data, dg.ctypes.data, ctypes.addressof(nf))
    return info, x, D, dg, nf.value
