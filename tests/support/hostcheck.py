"""ctypes wrapper over tests/support/libslpx_hostcheck.so — TEST INFRASTRUCTURE ONLY.

Sequential host interpreter of the compiled device plans (see hostcheck.cpp).  Lets
the CPU-only test tier validate the tape compiler, KKT plan and symbolic LDLᵀ
against the oracle.  Never used by the product.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

import sleipnir_amd

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libslpx_hostcheck.so"


def build():
    sleipnir_amd.build()
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++23", "-fPIC", "-shared", "--offload-arch=gfx950",
           "-x", "hip", str(HERE / "hostcheck.cpp"), "-o", str(LIB_PATH),
           "-L" + str(sleipnir_amd.LIB_PATH.parent), "-lslpx",
           "-Wl,-rpath," + str(sleipnir_amd.LIB_PATH.parent)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building hostcheck failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    sleipnir_amd.lib()  # make sure libslpx.so is loaded first (same arena)
    src = HERE / "hostcheck.cpp"
    if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime or \
            LIB_PATH.stat().st_mtime < sleipnir_amd.LIB_PATH.stat().st_mtime:
        build()
    L = ctypes.CDLL(str(LIB_PATH))
    vp, i32, d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double

    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("hc_create", vp, vp, vp, i32, i32, i32)
    sig("hc_destroy", None, vp)
    sig("hc_plan_hash", None, vp, vp)
    sig("hc_is_dense", i32, vp)
    sig("hc_mf_level_errors", i32, vp, d, d, vp, i32)
    sig("hc_tape_jit_compiles", i32, vp)
    sig("hc_supernodes", None, vp, vp)
    sig("hc_ldlt_tree", None, vp, vp, vp)
    sig("hc_supernode_plan", None, vp, vp, i32)
    sig("hc_mf_plan", None, vp, vp)
    sig("hc_info", None, vp, vp)
    sig("hc_mf_fronts", i32, vp, vp, i32)
    sig("hc_pattern", i32, vp, ctypes.c_int, vp, vp)
    sig("hc_perm", None, vp, vp)
    sig("hc_set_scaling", None, vp, vp)
    sig("hc_sweep", None, vp, vp, vp, vp, ctypes.c_int, vp)
    sig("hc_assemble", None, vp, vp, vp, vp)
    sig("hc_set_lhs", None, vp, vp)
    sig("hc_rhs", None, vp, vp, vp, vp, d, vp)
    sig("hc_set_rhs", None, vp, vp)
    sig("hc_factor", None, vp, d, d, vp, vp)
    sig("hc_solve", None, vp, vp)
    sig("hc_solve_after_factor", None, vp, vp)
    sig("hc_backsub", None, vp, vp, vp, d, vp, vp)
    _lib = L
    return L


INFO_KEYS = ["n", "m_e", "m_i", "nV", "nnz_lhs", "nnz_L", "ldlt_rounds", "ldlt_tasks",
             "etree_height", "ldlt_pairs", "tape_tasks", "tape_nodes", "tape_slots", "tape_edges",
             "tape_levels", "tape_slot_levels", "struct_singular", "off_g", "off_Ae", "off_Ai",
             "off_Hf", "off_Hc", "tape_global_tasks", "tape_large_tasks", "tape_shared_tasks",
             "tape_structure_bytes"]


def _fa(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HostCheck:
    def __init__(self, problem: "sleipnir_amd.Problem", perm=None, task_entries=0, small_lds_bytes=0):
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        self._h = lib().hc_create(problem._h, None if p is None else p.ctypes.data,
                                  0 if p is None else len(p), task_entries, small_lds_bytes)
        out = np.zeros(len(INFO_KEYS) + 4, dtype=np.int64)
        lib().hc_info(self._h, out.ctypes.data)
        self.info = {k: int(out[i]) for i, k in enumerate(INFO_KEYS)}
        self.n, self.m_e, self.m_i = self.info["n"], self.info["m_e"], self.info["m_i"]
        self.dense = bool(lib().hc_is_dense(self._h))
        self.info["ldlt_dense_pivoted"] = int(lib().hc_is_dense(self._h) == 2)

    def close(self):
        if self._h:
            lib().hc_destroy(self._h)
            self._h = None

    def mf_level_errors(self, delta, gamma):
        """The multifrontal plan in double against long double, level by level (hostcheck.cpp: hc_mf_level_errors):
        rows of {phase, round, level, values, worst normwise error of a front, median and max componentwise error}.
        The system is the one set by assemble() / rhs() (or set_lhs / set_rhs)."""
        out = np.zeros((4096, 7))
        rows = lib().hc_mf_level_errors(self._h, float(delta), float(gamma), out.ctypes.data, 4096)
        if rows < 0:
            raise RuntimeError("no multifrontal plan (SLPX_LDLT_MF=1)")
        return out[:rows]

    def is_dense(self):
        """the system is factored as a dense matrix (no column of L fits a task of the sparse plan, or SLPX_DENSE=1)"""
        return bool(lib().hc_is_dense(self._h))

    def plan_hash(self):
        """FNV-1a of [structure, full tape, values tape, KKT plan, LDLT plan, multifrontal plan]"""
        out = np.zeros(6, dtype=np.uint64)
        lib().hc_plan_hash(self._h, out.ctypes.data)
        return [int(v) for v in out]

    def tape_jit_compiles(self):
        """Bodies of the run-time generated tape kernel that hipRTC compiled for gfx950 (-1: rejected)."""
        return lib().hc_tape_jit_compiles(self._h)

    def supernodes(self):
        """Fundamental supernodes of L: count, widest, longest column, columns in >=4 / >=16 wide ones."""
        out = np.zeros(5, dtype=np.int64)
        lib().hc_supernodes(self._h, out.ctypes.data)
        return dict(zip(("count", "widest", "longest_column", "cols_in_ge4", "cols_in_ge16"), (int(v) for v in out)))

    def supernode_plan(self):
        """Supernodes the numeric kernels work on (relaxed: equal structure, any number of children)."""
        out = np.zeros(3 + 40, dtype=np.int64)
        lib().hc_supernode_plan(self._h, out.ctypes.data, len(out))
        return {"count": int(out[0]), "widest": int(out[1]), "critical_levels": int(out[2]),
                "width_hist": {w: int(c) for w, c in enumerate(out[3:]) if c}}

    def mf_plan(self):
        """The multifrontal plan (SLPX_LDLT_MF=1 when the handle was made), or {"built": False}."""
        out = np.zeros(10, dtype=np.int64)
        lib().hc_mf_plan(self._h, out.ctypes.data)
        keys = ("built", "fronts", "mfma_fronts", "max_children_values", "widest", "most_rows", "nnz_L",
                "table_bytes", "arena_doubles", "update_slots")
        d = dict(zip(keys, (int(v) for v in out)))
        d["built"] = bool(d["built"])
        return d

    def mf_fronts(self):
        """Every front of the multifrontal plan: rows of (task, round, level, w, nr, nch, n_s, flags)."""
        n = lib().hc_mf_fronts(self._h, None, 0)
        out = np.zeros((max(n, 1), 8), dtype=np.int32)
        lib().hc_mf_fronts(self._h, out.ctypes.data, n)
        return out[:n]

    def ldlt_tree(self):
        """(parent, column count) of the elimination tree in the permuted space."""
        dim = self.n + self.m_e
        parent = np.zeros(dim, dtype=np.int32)
        cc = np.zeros(dim, dtype=np.int32)
        lib().hc_ldlt_tree(self._h, parent.ctypes.data, cc.ctypes.data)
        return parent, cc

    def pattern(self, which):
        nnz = lib().hc_pattern(self._h, which, None, None)
        ncols = self.n if which != 5 else self.n + self.m_e
        colptr = np.zeros(ncols + 1, dtype=np.int32)
        rowidx = np.zeros(max(nnz, 1), dtype=np.int32)
        lib().hc_pattern(self._h, which, colptr.ctypes.data, rowidx.ctypes.data)
        return colptr, rowidx[:nnz]

    def perm(self):
        p = np.zeros(self.n + self.m_e, dtype=np.int32)
        lib().hc_perm(self._h, p.ctypes.data)
        return p

    def set_scaling(self, scales):
        s = _fa(scales)
        lib().hc_set_scaling(self._h, s.ctypes.data)

    def sweep(self, x, y=None, z=None, full=True):
        x = _fa(x)
        y = _fa(np.zeros(self.m_e) if y is None else y)
        z = _fa(np.zeros(self.m_i) if z is None else z)
        V = np.zeros(self.info["nV"])
        lib().hc_sweep(self._h, x.ctypes.data, y.ctypes.data, z.ctypes.data, int(full), V.ctypes.data)
        return V

    def assemble(self, s, z):
        s, z = _fa(s), _fa(z)
        lhs = np.zeros(self.info["nnz_lhs"])
        lib().hc_assemble(self._h, s.ctypes.data, z.ctypes.data, lhs.ctypes.data)
        return lhs

    def set_lhs(self, lhs):
        lhs = _fa(lhs)
        lib().hc_set_lhs(self._h, lhs.ctypes.data)

    def rhs(self, s, y, z, mu):
        s, y, z = _fa(s), _fa(y), _fa(z)
        out = np.zeros(self.n + self.m_e)
        lib().hc_rhs(self._h, s.ctypes.data, y.ctypes.data, z.ctypes.data, mu, out.ctypes.data)
        return out

    def set_rhs(self, rhs):
        rhs = _fa(rhs)
        lib().hc_set_rhs(self._h, rhs.ctypes.data)

    def factor(self, delta, gamma):
        D = np.zeros(self.n + self.m_e)
        stats = np.zeros(5)
        lib().hc_factor(self._h, delta, gamma, D.ctypes.data, stats.ctypes.data)
        return D, stats

    def solve(self):
        p = np.zeros(self.n + self.m_e)
        lib().hc_solve(self._h, p.ctypes.data)
        return p

    def solve_after_factor(self):
        """Backward substitution on the z the factorization produced (rhs carried as a row)."""
        p = np.zeros(self.n + self.m_e)
        lib().hc_solve_after_factor(self._h, p.ctypes.data)
        return p

    def backsub(self, s, z, mu):
        s, z = _fa(s), _fa(z)
        ps, pz = np.zeros(max(self.m_i, 1)), np.zeros(max(self.m_i, 1))
        lib().hc_backsub(self._h, s.ctypes.data, z.ctypes.data, mu, ps.ctypes.data, pz.ctypes.data)
        return ps[:self.m_i], pz[:self.m_i]
