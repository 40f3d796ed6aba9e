"""The linear-solver seam on its own: slpx_ldlt_create / set_matrix / compute / solve
(include/slpx.h) against a dense numpy solve of the same regularized matrix.

Follows what the reference asks of RegularizedLDLT (util/regularized_ldlt.hpp:45-151): same
pattern on every compute(), inertia (n, m_e, 0) or regularization until it is, solve() reusable
after one compute().
"""
import os

import numpy as np
import pytest

import sleipnir_amd as sa

pytestmark = pytest.mark.gpu


def _kkt(rng, n, m_e, definite=True, density=0.15, c22=0.0):
    """Dense symmetric [[H, A^T], [A, 0]] and the CSC arrays of its lower triangle (no stored
    diagonal in the (2,2) block)."""
    M = rng.standard_normal((n, n)) * (rng.random((n, n)) < density)
    H = M @ M.T + (np.eye(n) if definite else -0.5 * np.eye(n))
    if not definite:
        H -= 2.0 * np.diag(rng.random(n) < 0.3)
    A = rng.standard_normal((m_e, n)) * (rng.random((m_e, n)) < 2 * density)
    A[np.arange(m_e), rng.permutation(n)[:m_e]] += 2.0  # full row rank
    K = np.block([[H, A.T], [A, -c22 * np.eye(m_e)]])
    colptr, rowidx, vals = [0], [], []
    for c in range(n + m_e):
        for r in range(c, n + m_e):
            if K[r, c] != 0.0:
                rowidx.append(r)
                vals.append(K[r, c])
        colptr.append(len(rowidx))
    return K, np.array(colptr, np.int32), np.array(rowidx, np.int32), np.array(vals)


def _check_solution(K, n, reg, rhs, x):
    Kreg = K + np.diag(np.r_[np.full(n, reg[0]), np.full(K.shape[0] - n, -reg[1])])
    ref = np.linalg.solve(Kreg, rhs)
    np.testing.assert_allclose(x, ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())
    eig = np.linalg.eigvalsh(Kreg)
    assert (eig > 0).sum() == n and (eig < 0).sum() == K.shape[0] - n


def test_quasidefinite_matrix_needs_no_regularization():
    rng = np.random.default_rng(5)
    n, m_e = 60, 25
    K, colptr, rowidx, vals = _kkt(rng, n, m_e, c22=1e-2)
    ls = sa.System.linear_solver(n, m_e, colptr, rowidx)
    ls.reset_regularization(1e-10)
    ls.set_matrix(vals)
    info, reg, nfact = ls.compute()
    assert info[0] == 0 and nfact == 1 and reg[0, 0] == 0.0 and reg[0, 1] == 0.0
    for _ in range(3):  # solve() is reusable after one compute()
        rhs = rng.standard_normal(n + m_e)
        ls.set_rhs(rhs)
        ls.solve()
        _check_solution(K, n, reg[0], rhs, ls.get("p")[0])
    ls.close()


def test_zero_constraint_block_gets_the_first_regularization():
    # With a zero (2,2) block the fill-reducing order may eliminate a constraint row first: an
    # exactly-zero pivot, which Eigen reports as NumericalIssue, so the reference goes on to
    # delta = 1e-4, gamma = gamma_min (regularized_ldlt.hpp:74-102) — and so does this.  The
    # pivot is then -gamma_min: a factorization without pivoting loses digits to that growth
    # (the reference's SimplicialLDLT has none either), hence the residual test.
    rng = np.random.default_rng(5)
    n, m_e = 60, 25
    K, colptr, rowidx, vals = _kkt(rng, n, m_e)
    ls = sa.System.linear_solver(n, m_e, colptr, rowidx)
    ls.reset_regularization(1e-10)
    ls.set_matrix(vals)
    info, reg, nfact = ls.compute()
    assert info[0] == 0
    assert tuple(reg[0]) in ((0.0, 0.0), (1e-4, 1e-10))
    rhs = rng.standard_normal(n + m_e)
    ls.set_rhs(rhs)
    ls.solve()
    x = ls.get("p")[0]
    Kreg = K + np.diag(np.r_[np.full(n, reg[0, 0]), np.full(m_e, -reg[0, 1])])
    assert np.abs(Kreg @ x - rhs).max() < 1e-3 * np.abs(rhs).max()
    ls.close()


def test_indefinite_hessian_is_regularized_to_the_ideal_inertia():
    rng = np.random.default_rng(11)
    n, m_e = 48, 12
    K, colptr, rowidx, vals = _kkt(rng, n, m_e, definite=False, c22=1e-3)
    assert (np.linalg.eigvalsh(K) > 0).sum() != n
    ls = sa.System.linear_solver(n, m_e, colptr, rowidx)
    ls.reset_regularization(1e-10)
    ls.set_matrix(vals)
    info, reg, nfact = ls.compute()
    assert info[0] == 0 and nfact > 1 and reg[0, 0] > 0.0
    rhs = rng.standard_normal(n + m_e)
    ls.set_rhs(rhs)
    ls.solve()
    _check_solution(K, n, reg[0], rhs, ls.get("p")[0])
    # same pattern, new values: the next compute() starts from the remembered delta
    # (regularized_ldlt.hpp:98-107) and still lands on a correct factorization
    vals2 = vals * (1.0 + 0.05 * rng.standard_normal(vals.size))
    K2 = np.zeros_like(K)
    k = 0
    for c in range(n + m_e):
        for p in range(colptr[c], colptr[c + 1]):
            K2[rowidx[p], c] = K2[c, rowidx[p]] = vals2[k]
            k += 1
    ls.set_matrix(vals2)
    info, reg, _ = ls.compute()
    assert info[0] == 0
    ls.set_rhs(rhs)
    ls.solve()
    _check_solution(K2, n, reg[0], rhs, ls.get("p")[0])
    ls.close()


@pytest.mark.parametrize("batch", [3, 200])  # 200: the batch-interleaved factorization
def test_batch_of_matrices_with_one_pattern(batch):
    rng = np.random.default_rng(23)
    n, m_e = 40, 16
    K, colptr, rowidx, vals = _kkt(rng, n, m_e, c22=1e-2)
    scale = 1.0 + 0.1 * rng.random((batch, 1))
    ls = sa.System.linear_solver(n, m_e, colptr, rowidx, batch=batch)
    ls.reset_regularization(1e-10)
    ls.set_matrix(vals[None, :] * scale)
    info, reg, _ = ls.compute()
    assert (info == 0).all()
    rhs = rng.standard_normal((batch, n + m_e))
    ls.set_rhs(rhs)
    ls.solve()
    x = ls.get("p")
    for b in (0, batch // 2, batch - 1):
        _check_solution(K * scale[b, 0], n, reg[b], rhs[b], x[b])
    ls.close()


def test_pattern_must_be_a_lower_triangle():
    with pytest.raises(sa.SlpxError):
        sa.System.linear_solver(2, 0, [0, 1, 3], [0, 0, 1])


def _structured_kkt(rng, kind, n, m_e):
    """Quasi-definite test matrices with the structures that make supernodes: banded (chains of
    equal structure), block-arrow (a dense border: wide trapezoids), random sparse."""
    dim = n + m_e
    H = np.zeros((n, n))
    if kind == "banded":
        bw = int(rng.integers(2, 9))
        for d in range(1, bw + 1):
            v = rng.standard_normal(n - d) * 0.3
            H += np.diag(v, -d) + np.diag(v, d)
    elif kind == "arrow":
        blk = int(rng.integers(3, 9))
        for b0 in range(0, n - blk, blk):
            B = rng.standard_normal((blk, blk)) * 0.3
            H[b0:b0 + blk, b0:b0 + blk] += B + B.T
        border = int(rng.integers(2, 12))
        W = rng.standard_normal((border, n)) * (rng.random((border, n)) < 0.3) * 0.3
        H[n - border:, :] += W
        H[:, n - border:] += W.T
    else:
        M = rng.standard_normal((n, n)) * (rng.random((n, n)) < 1.5 / n) * 0.3
        H = M + M.T
    H += np.diag(np.abs(H).sum(axis=1) + 1.0)  # strictly diagonally dominant: positive definite
    # constraint rows couple a few NEIGHBOURING variables, like the rows of a transcription (a
    # dense A makes L dense, which is the reference's dense-LDLT territory)
    A = np.zeros((m_e, n))
    centres = np.sort(rng.integers(0, n, size=m_e))
    for r, c0 in enumerate(centres):
        cols = np.unique(np.clip(c0 + rng.integers(-4, 5, size=3), 0, n - 1))
        A[r, cols] = rng.standard_normal(len(cols))
    # full row rank without long-range couplings: every row gets a pivot column of its own nearby
    free = np.ones(n, dtype=bool)
    for r, c0 in enumerate(centres):
        c = next(int(k) for k in sorted(range(n), key=lambda k: abs(k - int(c0))) if free[k])
        free[c] = False
        A[r, c] += 2.0 + abs(A[r]).sum()
    K = np.block([[H, A.T], [A, -1e-2 * np.eye(m_e)]])
    colptr, rowidx, vals = [0], [], []
    for c in range(dim):
        for r in range(c, dim):
            if K[r, c] != 0.0:
                rowidx.append(r)
                vals.append(K[r, c])
        colptr.append(len(rowidx))
    return K, np.array(colptr, np.int32), np.array(rowidx, np.int32), np.array(vals)


@pytest.mark.parametrize("kind", ["banded", "arrow", "random"])
def test_irregular_patterns_against_dense_solves(kind):
    """Twelve seeded matrices per structure, 40 to 600 unknowns: supernode chains of every width,
    trapezoids up to a wave of rows, one to several rounds of tasks, hand-overs both ways — the
    factorization with the right-hand side riding in it (compute + solve after a step-like
    sequence) and the forward / backward pair for a new right-hand side, each against numpy."""
    rng = np.random.default_rng({"banded": 11, "arrow": 12, "random": 13}[kind])
    seen_chain = False
    for trial in range(12):
        # (a random sparse graph is an expander: its factor fills in almost completely whatever the
        # ordering, so those stay small enough for a column to fit a task — DESIGN.md §6)
        n = int(rng.integers(30, 110 if kind == "random" else 450))
        m_e = int(rng.integers(5, max(6, n // 3)))
        K, colptr, rowidx, vals = _structured_kkt(rng, kind, n, m_e)
        ls = sa.System.linear_solver(n, m_e, colptr, rowidx)
        seen_chain = seen_chain or ls.info["ldlt_widest_supernode"] >= 2
        ls.reset_regularization(1e-10)
        ls.set_matrix(vals)
        info, reg, nfact = ls.compute()
        assert info[0] == 0 and nfact == 1 and reg[0, 0] == 0.0, (kind, trial, n, m_e, info, reg)
        for _ in range(2):
            rhs = rng.standard_normal(n + m_e)
            ls.set_rhs(rhs)
            ls.solve()
            _check_solution(K, n, reg[0], rhs, ls.get("p")[0])
        ls.close()
    assert seen_chain or kind == "random"


def test_compute_before_any_right_hand_side_on_recycled_memory(monkeypatch):
    """RegularizedLDLT::compute(lhs) may come before any solve(rhs) (regularized_ldlt.hpp:72,87).  The
    right-hand side rides in the factorization as an extra row, so its buffer must not start as
    whatever the allocator recycles: memory of an earlier solver can hold the hand-over sentinel
    (a NaN pattern), a NaN keeps its payload through arithmetic, and an update block that IS the
    sentinel is never taken.  Seen on the pair-list kernels (SLPX_LDLT_MF=0: a launch per phase): the fourth solver
    of this very sequence spun to its time-out 26 times and reported NumericalIssue."""
    monkeypatch.setenv("SLPX_LDLT_MF", "0")
    rng = np.random.default_rng(12)
    for trial in range(5):
        n = int(rng.integers(30, 450))
        m_e = int(rng.integers(5, max(6, n // 3)))
        K, colptr, rowidx, vals = _structured_kkt(rng, "arrow", n, m_e)
        ls = sa.System.linear_solver(n, m_e, colptr, rowidx)
        ls.reset_regularization(1e-10)
        ls.set_matrix(vals)
        info, reg, nfact = ls.compute()  # no set_rhs before it
        assert info[0] == 0 and nfact == 1, (trial, info, reg, nfact)
        for _ in range(2):  # (the generator's sequence of the test above)
            rhs = rng.standard_normal(n + m_e)
            ls.set_rhs(rhs)
            ls.solve()
            _check_solution(K, n, reg[0], rhs, ls.get("p")[0])
        ls.close()
