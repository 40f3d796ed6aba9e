"""sleipnir_amd — MI355X-native interior-point Newton step behind Sleipnir's
``slp::Problem`` surface.

The product is the C++/HIP library ``libslpx.so`` (sources in ``csrc/``, C-ABI in
``include/slpx.h``).  This Python package is plumbing only: it builds the library
in-tree and exposes the C-ABI through ctypes so that ``tests/`` and ``bench.py``
can drive it.  It never computes anything itself and has no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent
# (SLPX_LIB: another build of the library, e.g. for an A/B on one box)
LIB_PATH = Path(os.environ["SLPX_LIB"]).resolve() if os.environ.get("SLPX_LIB") else _PKG / "libslpx.so"

ABI_VERSION = 6  # include/slpx.h: SLPX_ABI_VERSION (struct layouts and entry points below)

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)

# include/slpx.h: SLPX_INFO_*
INFO_KEYS = [
    "n", "m_e", "m_i", "nV", "nnz_g", "nnz_Ae", "nnz_Ai", "nnz_Hf", "nnz_Hc", "nnz_lhs", "nnz_L",
    "ldlt_rounds", "ldlt_tasks", "etree_height", "ldlt_pairs", "tape_tasks", "tape_nodes",
    "tape_slots", "tape_edges", "tape_levels", "tape_slot_levels", "assemble_bytes", "rhs_bytes",
    "factor_bytes", "solve_bytes", "sweep_bytes", "struct_singular", "off_g", "off_Ae", "off_Ai",
    "off_Hf", "off_Hc", "graph_nodes", "nonlinear_rows", "tape_global_tasks", "tape_shared_tasks",
    "tape_program_bytes", "ldlt_levels", "ldlt_supernodes", "ldlt_widest_supernode",
    "ldlt_multifrontal", "ldlt_fronts", "ldlt_mfma_fronts", "ldlt_dense",
]

# slpx_op
OPS = {name: i for i, name in enumerate([
    "CONST", "VAR", "ADD", "SUB", "NEG", "MUL", "DIV", "POW", "ABS", "SIGN", "SQRT", "CBRT", "EXP",
    "LOG", "LOG10", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "ATAN2", "SINH", "COSH", "TANH",
    "ERF", "HYPOT", "MAX", "MIN", "ISNONNEG", "ISPOS"])}


class Options(ctypes.Structure):
    _fields_ = [("tolerance", ctypes.c_double), ("max_iterations", ctypes.c_int32),
                ("timeout", ctypes.c_double), ("feasible_ipm", ctypes.c_int32),
                ("diagnostics", ctypes.c_int32), ("spy", ctypes.c_int32)]


class Report(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("factorizations", ctypes.c_int32),
                ("solves", ctypes.c_int32), ("value_sweeps", ctypes.c_int32),
                ("delta", ctypes.c_double), ("gamma", ctypes.c_double),
                ("final_error", ctypes.c_double), ("t_setup", ctypes.c_double),
                ("t_kkt_build", ctypes.c_double), ("t_kkt_decomp", ctypes.c_double),
                ("t_kkt_solve", ctypes.c_double), ("t_line_search", ctypes.c_double),
                ("t_ad_refresh", ctypes.c_double), ("t_total", ctypes.c_double),
                ("t_compile", ctypes.c_double), ("restorations", ctypes.c_int32),
                ("restoration_iterations", ctypes.c_int32), ("t_restoration_setup", ctypes.c_double),
                ("t_restoration", ctypes.c_double)]


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile libslpx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_PKG / "csrc"), "-j", str(os.cpu_count() or 4)]
    if force:
        subprocess.run(cmd + ["clean"], check=True, capture_output=not verbose)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libslpx.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """Load libslpx.so (building it first if missing) and declare the C-ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = ctypes.CDLL(str(LIB_PATH))
    vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double

    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("slpx_abi_version", ctypes.c_int)
    if L.slpx_abi_version() != ABI_VERSION:
        raise SlpxError(f"{LIB_PATH} has ABI version {L.slpx_abi_version()}, this binding was written for {ABI_VERSION}")
    sig("slpx_last_error", ctypes.c_char_p)
    sig("slpx_device_count", ctypes.c_int)
    sig("slpx_shard_range", ctypes.c_int, i64, i32, i32, c_i64p, c_i64p)
    sig("slpx_graph_reset", None)
    sig("slpx_graph_size", i64)
    sig("slpx_expr_variable", i32, f64)
    sig("slpx_expr_constant", i32, f64)
    sig("slpx_expr_unary", i32, ctypes.c_int, i32)
    sig("slpx_expr_binary", i32, ctypes.c_int, i32, i32)
    sig("slpx_expr_type", ctypes.c_int, i32)
    sig("slpx_expr_value", f64, i32)
    sig("slpx_expr_set_value", None, i32, f64)
    sig("slpx_expr_gradient_tree", None, i32, vp, i32, vp)
    sig("slpx_problem_create", vp)
    sig("slpx_problem_destroy", None, vp)
    sig("slpx_problem_decision_variable", i32, vp)
    sig("slpx_problem_minimize", None, vp, i32)
    sig("slpx_problem_maximize", None, vp, i32)
    sig("slpx_problem_subject_to_eq", None, vp, i32)
    sig("slpx_problem_subject_to_ineq", None, vp, i32)
    sig("slpx_problem_cost_type", ctypes.c_int, vp)
    sig("slpx_problem_eq_type", ctypes.c_int, vp)
    sig("slpx_problem_ineq_type", ctypes.c_int, vp)
    sig("slpx_problem_dims", None, vp, c_i32p, c_i32p, c_i32p)
    sig("slpx_problem_get_x", None, vp, vp)
    sig("slpx_problem_set_x", None, vp, vp)
    sig("slpx_problem_solve", ctypes.c_int, vp, ctypes.POINTER(Options), ctypes.POINTER(Report))
    sig("slpx_problem_solve_sized", ctypes.c_int, vp, ctypes.POINTER(Options), ctypes.c_uint32, ctypes.POINTER(Report))
    sig("slpx_problem_get_duals", None, vp, vp, vp, vp)
    sig("slpx_problem_restoration_steps", ctypes.c_int, vp, ctypes.POINTER(Options), vp, vp, vp, vp, f64, i32)
    sig("slpx_problem_prebuild_kernels", ctypes.c_int, vp, ctypes.c_char_p)
    sig("slpx_system_create", vp, vp, i32, i32, vp, i32)
    sig("slpx_system_destroy", None, vp)
    sig("slpx_system_set_stream", ctypes.c_int, vp, vp)
    sig("slpx_system_sync", ctypes.c_int, vp)
    sig("slpx_system_info", ctypes.c_int, vp, vp)
    sig("slpx_system_pattern", i32, vp, ctypes.c_int, vp, vp)
    sig("slpx_system_perm", ctypes.c_int, vp, vp)
    sig("slpx_system_set_scaling", ctypes.c_int, vp, vp)
    sig("slpx_system_set_state", ctypes.c_int, vp, vp, vp, vp, vp, vp)
    sig("slpx_tape_sweep", ctypes.c_int, vp, ctypes.c_int)
    sig("slpx_kkt_assemble", ctypes.c_int, vp)
    sig("slpx_kkt_rhs", ctypes.c_int, vp)
    sig("slpx_ldlt_factor", ctypes.c_int, vp, vp, vp, vp)
    sig("slpx_ldlt_compute", ctypes.c_int, vp, vp, vp, vp)
    sig("slpx_ldlt_reset", ctypes.c_int, vp, f64)
    sig("slpx_ldlt_solve", ctypes.c_int, vp)
    sig("slpx_step_backsub", ctypes.c_int, vp)
    sig("slpx_newton_step", ctypes.c_int, vp, ctypes.c_int, vp)
    sig("slpx_system_get", i64, vp, ctypes.c_int, vp)
    sig("slpx_system_set_rhs", ctypes.c_int, vp, vp)
    sig("slpx_system_set_lhs", ctypes.c_int, vp, vp)
    sig("slpx_system_time_step", ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp)
    sig("slpx_debug_chain", ctypes.c_int, vp, ctypes.c_int)
    sig("slpx_system_time_fused_step", ctypes.c_int, vp, ctypes.c_int, vp)
    sig("slpx_newton_steps", ctypes.c_int, vp, i32, ctypes.c_int, ctypes.c_int, vp)
    sig("slpx_system_regularization", ctypes.c_int, vp, vp)
    sig("slpx_problem_add_callback", ctypes.c_int, vp, vp, vp)
    sig("slpx_problem_clear_callbacks", ctypes.c_int, vp)
    sig("slpx_problem_system", vp, vp)
    sig("slpx_ldlt_create", vp, i32, i32, vp, vp, i32, i32)
    sig("slpx_ldlt_set_matrix", ctypes.c_int, vp, vp)
    sig("slpx_ipm_direction", ctypes.c_int, vp, f64, vp)
    sig("slpx_ipm_trial", ctypes.c_int, vp, f64, ctypes.c_int, vp)
    sig("slpx_ipm_commit", ctypes.c_int, vp, f64, f64, ctypes.c_int)
    sig("slpx_ipm_errors", ctypes.c_int, vp, vp, vp)
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class SlpxError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise SlpxError(lib().slpx_last_error().decode())
    return rc


class IterationInfo(ctypes.Structure):
    _fields_ = [("iteration", ctypes.c_int32), ("n", ctypes.c_int32), ("m_e", ctypes.c_int32),
                ("m_i", ctypes.c_int32)] + [(k, ctypes.POINTER(ctypes.c_double)) for k in "xsyzV"] + [
                    ("off", ctypes.c_int64 * 8), ("in_restoration", ctypes.c_int32)]


IterationCallback = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(IterationInfo), ctypes.c_void_p)


class Problem:
    """Handle on an slp::Problem living in libslpx (include/slpx.h, slpx_problem_*)."""

    def __init__(self, handle=None):
        self._h = handle if handle is not None else lib().slpx_problem_create()
        if not self._h:
            raise SlpxError(lib().slpx_last_error().decode())

    def close(self):
        if self._h:
            lib().slpx_problem_destroy(self._h)
            self._h = None

    @property
    def dims(self):
        n, me, mi = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        lib().slpx_problem_dims(self._h, ctypes.byref(n), ctypes.byref(me), ctypes.byref(mi))
        return n.value, me.value, mi.value

    def decision_variable(self) -> int:
        return lib().slpx_problem_decision_variable(self._h)

    def minimize(self, expr_id: int):
        lib().slpx_problem_minimize(self._h, expr_id)

    def maximize(self, expr_id: int):
        lib().slpx_problem_maximize(self._h, expr_id)

    def subject_to_eq(self, expr_id: int):
        lib().slpx_problem_subject_to_eq(self._h, expr_id)

    def subject_to_ineq(self, expr_id: int):
        lib().slpx_problem_subject_to_ineq(self._h, expr_id)

    def types(self):
        L = lib()
        return (L.slpx_problem_cost_type(self._h), L.slpx_problem_eq_type(self._h),
                L.slpx_problem_ineq_type(self._h))

    def get_x(self) -> np.ndarray:
        x = np.zeros(self.dims[0])
        lib().slpx_problem_get_x(self._h, x.ctypes.data)
        return x

    def set_x(self, x):
        x = _f64(x)
        lib().slpx_problem_set_x(self._h, x.ctypes.data)

    def solve(self, tolerance=1e-8, max_iterations=5000, timeout=0.0, feasible_ipm=False,
              diagnostics=False, spy=False):
        opt = Options(tolerance, max_iterations, timeout, int(feasible_ipm), int(diagnostics), int(spy))
        rep = Report()
        status = lib().slpx_problem_solve_sized(self._h, ctypes.byref(opt), ctypes.sizeof(Options), ctypes.byref(rep))
        if status == -100:
            raise SlpxError(lib().slpx_last_error().decode())
        return status, {f[0]: getattr(rep, f[0]) for f in Report._fields_}

    def restoration_steps(self, x, s, y, z, mu, steps, tolerance=1e-8, max_iterations=5000):
        """feasibility_restoration from the given iterate, `steps` iterations (slpx_problem_restoration_steps)."""
        x, s, y, z = (np.array(a, dtype=np.float64, copy=True) for a in (x, s, y, z))
        opt = Options(tolerance, max_iterations, 0.0, 0, 0, 0)
        status = lib().slpx_problem_restoration_steps(self._h, ctypes.byref(opt), x.ctypes.data, s.ctypes.data,
                                                      y.ctypes.data, z.ctypes.data, float(mu), int(steps))
        if status == -100:
            raise SlpxError(lib().slpx_last_error().decode())
        return status, x, s, y, z

    def add_callback(self, fn):
        """Problem::add_callback (problem.hpp:690-709): fn(info: dict) -> truthy to stop."""

        def trampoline(info_ptr, _user):
            i = info_ptr.contents
            view = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy() if k else np.zeros(0)
            off = list(i.off)
            return int(bool(fn({"iteration": i.iteration, "x": view(i.x, i.n), "s": view(i.s, i.m_i),
                                "y": view(i.y, i.m_e), "z": view(i.z, i.m_i), "f": i.V[off[0]], "off": off,
                                "V": i.V, "in_restoration": bool(i.in_restoration)})))

        cb = IterationCallback(trampoline)
        self._callbacks = getattr(self, "_callbacks", []) + [cb]  # keep the thunks alive
        _check(lib().slpx_problem_add_callback(self._h, ctypes.cast(cb, ctypes.c_void_p), None))

    def prebuild_kernels(self, directory=None) -> int:
        """Cross-compile this model's generated tape kernels for gfx950 into the library's
        jit_cache (slpx_problem_prebuild_kernels; needs no device)."""
        n = lib().slpx_problem_prebuild_kernels(self._h, None if directory is None else str(directory).encode())
        if n < 0:
            raise SlpxError(lib().slpx_last_error().decode())
        return n

    def system(self) -> "System":
        """The system solve() runs on (slpx_problem_system); owned by the problem."""
        h = lib().slpx_problem_system(self._h)
        if not h:
            raise SlpxError(lib().slpx_last_error().decode())
        return System._adopt(h, batch=1, borrowed=True)

    def clear_callbacks(self):
        _check(lib().slpx_problem_clear_callbacks(self._h))
        self._callbacks = []

    def duals(self):
        n, me, mi = self.dims
        s, y, z = np.zeros(mi), np.zeros(me), np.zeros(mi)
        lib().slpx_problem_get_duals(self._h, s.ctypes.data, y.ctypes.data, z.ctypes.data)
        return s, y, z


class System:
    """Compiled Newton system on one GPU (include/slpx.h, slpx_system_* and kernels)."""

    _borrowed = False

    @classmethod
    def _adopt(cls, handle, batch, borrowed=False):
        self = cls.__new__(cls)
        self._h = handle
        self._borrowed = borrowed
        self.batch = batch
        out = np.zeros(len(INFO_KEYS) + 4, dtype=np.int64)
        _check(lib().slpx_system_info(self._h, out.ctypes.data))
        self.info = {k: int(out[i]) for i, k in enumerate(INFO_KEYS)}
        return self

    @classmethod
    def linear_solver(cls, n, m_e, colptr, rowidx, batch=1, device=0):
        """RegularizedLDLT on its own (slpx_ldlt_create): lower-triangular CSC pattern only."""
        cp = np.ascontiguousarray(colptr, dtype=np.int32)
        ri = np.ascontiguousarray(rowidx, dtype=np.int32)
        h = lib().slpx_ldlt_create(int(n), int(m_e), cp.ctypes.data, ri.ctypes.data, int(batch), int(device))
        if not h:
            raise SlpxError(lib().slpx_last_error().decode())
        return cls._adopt(h, batch)

    def set_matrix(self, values):
        a = _f64(values)
        _check(lib().slpx_ldlt_set_matrix(self._h, a.ctypes.data))

    def __init__(self, problem: Problem, batch: int = 1, device: int = 0, perm=None):
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        self._h = lib().slpx_system_create(problem._h, batch, device, _ptr(p), 0 if p is None else len(p))
        if not self._h:
            raise SlpxError(lib().slpx_last_error().decode())
        self.batch = batch
        out = np.zeros(len(INFO_KEYS) + 4, dtype=np.int64)
        _check(lib().slpx_system_info(self._h, out.ctypes.data))
        self.info = {k: int(out[i]) for i, k in enumerate(INFO_KEYS)}

    def close(self):
        if self._h and not self._borrowed:
            lib().slpx_system_destroy(self._h)
        self._h = None

    def pattern(self, which: int):
        nnz = lib().slpx_system_pattern(self._h, which, None, None)
        ncols = self.info["n"] if which != 5 else self.info["n"] + self.info["m_e"]
        colptr = np.zeros(ncols + 1, dtype=np.int32)
        rowidx = np.zeros(max(nnz, 1), dtype=np.int32)
        lib().slpx_system_pattern(self._h, which, colptr.ctypes.data, rowidx.ctypes.data)
        return colptr, rowidx[:nnz]

    def perm(self):
        p = np.zeros(self.info["n"] + self.info["m_e"], dtype=np.int32)
        lib().slpx_system_perm(self._h, p.ctypes.data)
        return p

    def set_stream(self, stream_ptr: int):
        _check(lib().slpx_system_set_stream(self._h, stream_ptr))

    def sync(self):
        _check(lib().slpx_system_sync(self._h))

    def set_scaling(self, scales):
        s = _f64(scales)
        _check(lib().slpx_system_set_scaling(self._h, s.ctypes.data))

    def set_state(self, x=None, s=None, y=None, z=None, mu=None):
        x, s, y, z, mu = (_f64(a) for a in (x, s, y, z, mu))
        _check(lib().slpx_system_set_state(self._h, _ptr(x), _ptr(s), _ptr(y), _ptr(z), _ptr(mu)))

    def sweep(self, full=True):
        _check(lib().slpx_tape_sweep(self._h, int(full)))

    def assemble(self):
        _check(lib().slpx_kkt_assemble(self._h))

    def rhs(self):
        _check(lib().slpx_kkt_rhs(self._h))

    def factor(self, delta, gamma):
        d = np.full(self.batch, delta, dtype=np.float64) if np.isscalar(delta) else _f64(delta)
        g = np.full(self.batch, gamma, dtype=np.float64) if np.isscalar(gamma) else _f64(gamma)
        stats = np.zeros((self.batch, 5))
        _check(lib().slpx_ldlt_factor(self._h, d.ctypes.data, g.ctypes.data, stats.ctypes.data))
        return stats

    def compute(self):
        info = np.zeros(self.batch, dtype=np.int32)
        reg = np.zeros((self.batch, 2))
        nf = ctypes.c_int32()
        _check(lib().slpx_ldlt_compute(self._h, info.ctypes.data, reg.ctypes.data, ctypes.addressof(nf)))
        return info, reg, nf.value

    def reset_regularization(self, gamma_min=1e-10):
        _check(lib().slpx_ldlt_reset(self._h, gamma_min))

    def solve(self):
        _check(lib().slpx_ldlt_solve(self._h))

    def backsub(self):
        _check(lib().slpx_step_backsub(self._h))

    def newton_step(self, refresh_ad=True):
        info = np.zeros(self.batch, dtype=np.int32)
        _check(lib().slpx_newton_step(self._h, int(refresh_ad), info.ctypes.data))
        return info

    def newton_steps(self, count, refresh_ad=True, forget_regularization=True):
        info = np.zeros(self.batch, dtype=np.int32)
        _check(lib().slpx_newton_steps(self._h, int(count), int(refresh_ad), int(forget_regularization),
                                       info.ctypes.data))
        return info

    def regularization(self) -> np.ndarray:
        """[batch, 2] = (delta, gamma) the last compute / Newton step settled on."""
        reg = np.zeros((self.batch, 2))
        _check(lib().slpx_system_regularization(self._h, reg.ctypes.data))
        return reg

    def get(self, which: str) -> np.ndarray:
        sel = {"V": 0, "lhs": 1, "rhs": 2, "p": 3, "p_s": 4, "p_z": 5, "D": 6, "Lx": 7, "x": 8, "s": 9,
               "y": 10, "z": 11}[which]
        count = _check(lib().slpx_system_get(self._h, sel, None))
        out = np.zeros(max(count, 1))
        _check(lib().slpx_system_get(self._h, sel, out.ctypes.data))
        return out[:count].reshape(self.batch, -1)

    # ---- the interior-point iteration around the Newton step, on the resident iterate ----
    IPM_ERROR_KEYS = ["dual_inf_u", "sz_max_u", "ce_inf_u", "cis_inf_u", "y1_u", "z1_u", "dual_inf", "sz_min",
                      "sz_max", "ce_inf", "cis_inf", "y1", "z1", "f", "viol", "logsum", "aetce_sq", "ce_sq",
                      "aitcp_sq", "cp_sq", "x_inf", "s_inf", "finite", "ci_all_pos"]

    def ipm_direction(self, tau):
        out = np.zeros(3)
        _check(lib().slpx_ipm_direction(self._h, float(tau), out.ctypes.data))
        return {"alpha_max": out[0], "alpha_z": out[1], "D_phi": out[2]}

    def ipm_trial(self, alpha, s_from_ci=False):
        out = np.zeros(4)
        _check(lib().slpx_ipm_trial(self._h, float(alpha), int(s_from_ci), out.ctypes.data))
        return {"f": out[0], "viol": out[1], "logsum": out[2], "finite": out[3]}

    def ipm_commit(self, alpha, alpha_z, s_from_ci=False):
        _check(lib().slpx_ipm_commit(self._h, float(alpha), float(alpha_z), int(s_from_ci)))

    def ipm_errors(self, error_scales):
        sc = _f64(error_scales)
        out = np.zeros(24)
        _check(lib().slpx_ipm_errors(self._h, sc.ctypes.data, out.ctypes.data))
        return dict(zip(self.IPM_ERROR_KEYS, out))

    def set_rhs(self, rhs):
        r = _f64(rhs)
        _check(lib().slpx_system_set_rhs(self._h, r.ctypes.data))

    def set_lhs(self, lhs):
        a = _f64(lhs)
        _check(lib().slpx_system_set_lhs(self._h, a.ctypes.data))

    def debug_chain(self, action=0) -> int:
        """Chain failures recovered from so far; action 1 breaks the next chained sweep's hand-over (slpx_debug_chain)."""
        return _check(lib().slpx_debug_chain(self._h, action))

    def time_fused_step(self, iters=10):
        """The launches a single problem's step really makes (slpx_system_time_fused_step)."""
        ms = np.zeros(4, dtype=np.float32)
        _check(lib().slpx_system_time_fused_step(self._h, iters, ms.ctypes.data))
        return {"sweep": float(ms[0]), "kkt_factor_solve": float(ms[1]), "total": float(ms[2]),
                "one_launch": bool(ms[3]), "multifrontal": ms[3] == 2.0}

    def time_step(self, iters=10, refresh_ad=True):
        ms = np.zeros(8, dtype=np.float32)
        _check(lib().slpx_system_time_step(self._h, iters, int(refresh_ad), ms.ctypes.data))
        keys = ["sweep", "assemble", "rhs", "factor", "solve", "backsub", "total", "factorizations"]
        return {k: float(ms[i]) for i, k in enumerate(keys)}


from .dist import Comm, shard_range  # noqa: E402,F401  (multi-GPU plumbing)
