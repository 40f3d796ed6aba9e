"""The dense branch of the regularized LDLT (util/dense_regularized_ldlt.hpp:59-136; chosen by the reference where the
KKT system is dense, interior_point.hpp:340-352): a model whose Hessian is dense in hundreds of variables — single
shooting, optimization/ocp.hpp:382-400 — used to be REFUSED by the product ("a single column exceeds the LDS task
budget"); it is now factored as a dense matrix (ldlt_dense_kernels.h, LdltPlan::dense).  CPU tier: the plan and the
host interpreter of the dense factorization against the oracle; GPU tier: the kernels, and whole solves."""
import numpy as np
import pytest

from tests.support import model, parity, shooting


def build_both(N, **kw):
    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    return shooting.build(mo, N, **kw), shooting.build(mp, N, **kw)


@pytest.mark.parametrize("N,with_eq", [(300, False), (300, True)])
def test_single_shooting_takes_the_dense_plan(fresh, hostcheck, N, with_eq):
    """300 steps: a column of L has 300 entries, ~11 000 entry-equivalents against a task's 2048 (5120 at most)."""
    po, pp = build_both(N, final_state_constraint=with_eq)
    assert po.p.dims == pp.p.dims == (N, 1 if with_eq else 0, 2 * N)
    hc = hostcheck.HostCheck(pp.p)
    assert hc.is_dense()
    cp, ri = hc.pattern(5)
    assert len(ri) == (N + with_eq) * (N + with_eq + 1) // 2  # the lower triangle is full
    for case in ("step0", "interior"):
        errs = parity.check_newton_step(hc, po.p, case)
        assert errs["p"] <= 1e-8, errs


def test_the_plan_follows_the_references_rule(fresh, hostcheck, slpx, monkeypatch):
    """interior_point.hpp:340-352: sparse iff nnz(H) + nnz(tril A_i^T A_i) + nnz(A_e) < 0.25 (n + m_e)^2.  Single shooting
    over 40 steps has a full Hessian: 820 of 1600 — dense, with Eigen::LDLT's diagonal pivoting (VERDICT r05 missing 4;
    r05 took the dense branch only where the sparse plan was impossible).  The cart-pole transcription is sparse at every
    horizon.  SLPX_DENSE=0 keeps the sparse plan whatever the fill, SLPX_DENSE=1 asks for the plain (unpivoted) dense
    kernel — both give the oracle's step."""
    from tests.support import models

    po, pp = build_both(40)
    hc = hostcheck.HostCheck(pp.p)
    assert hc.is_dense() and hc.info["ldlt_dense_pivoted"] == 1
    parity.check_newton_step(hc, po.p, "interior")
    monkeypatch.setenv("SLPX_DENSE", "0")
    assert not hostcheck.HostCheck(pp.p).is_dense()
    monkeypatch.setenv("SLPX_DENSE", "1")
    hc = hostcheck.HostCheck(pp.p)
    assert hc.is_dense() and hc.info["ldlt_dense_pivoted"] == 0
    parity.check_newton_step(hc, po.p, "interior")
    monkeypatch.delenv("SLPX_DENSE")
    slpx.lib().slpx_graph_reset()
    for N in (2, 4, 100):
        cp = models.cart_pole(N, 5.0 / N)
        assert not hostcheck.HostCheck(cp).is_dense()
        cp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,with_eq", [(300, False), (300, True), (40, True)])
def test_dense_newton_step_gpu(fresh, slpx, monkeypatch, N, with_eq):
    """ldlt_dense_factor_kernel / ldlt_dense_solve_kernel under the oracle: lhs, rhs, inertia, D, p, p_s, p_z."""
    if N < 100:
        monkeypatch.setenv("SLPX_DENSE", "1")
    po, pp = build_both(N, final_state_constraint=with_eq)
    system = slpx.System(pp.p, batch=1, device=0)
    try:
        # (300 steps: dense by the reference's rule, pivoted; 40 steps under SLPX_DENSE=1: the plain dense kernel)
        assert system.info["ldlt_dense"] == (2 if N >= 100 else 1) and system.info["ldlt_tasks"] == 0
        for case in ("step0", "interior"):
            errs = parity.check_newton_step(parity.GpuBackend(system), po.p, case)
            assert errs["p"] <= 1e-8, errs
    finally:
        system.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,with_eq", [(300, False), (300, True)])
def test_single_shooting_solve_matches_the_oracle_gpu(fresh, slpx, N, with_eq):
    """The whole solve (interior point on the resident iterate, every factorization the dense kernel's): exit
    status, inputs and the states they give against the oracle's (which takes the reference's dense branch)."""
    po, pp = build_both(N, final_state_constraint=with_eq)
    so = po.solve()
    sp = pp.solve()
    assert sp == so == model.NlpProblem.SUCCESS
    uo, up = po.p.get_x(), pp.p.get_x()
    xo, xp = shooting.rollout(uo), shooting.rollout(up)
    assert np.max(np.abs(np.array(xo) - np.array(xp))) <= 1e-6
    assert np.max(np.abs(uo - up)) <= 1e-4 * shooting.U_MAX
    assert abs(xp[-1] - shooting.R) <= (1e-6 if with_eq else 1e-2)
    # bang, then hold: the first input at its bound, the last near the steady-state value r
    assert abs(up[0] - shooting.U_MAX) <= 1e-3 and abs(up[-2] - shooting.R) <= 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3, 70])
def test_linear_solver_seam_on_the_dense_branch(monkeypatch, batch):
    """slpx_ldlt_create / set_matrix / compute / solve (RegularizedLDLT with use_sparse = false,
    util/regularized_ldlt.hpp:45-87) on the dense kernels, single and batched (a workgroup per problem): a
    quasidefinite matrix needs no regularization, an indefinite Hessian is regularized to the ideal inertia — against
    dense numpy solves of the regularized matrices."""
    import sleipnir_amd as sa
    from tests.test_linear_solver_gpu import _check_solution, _kkt

    monkeypatch.setenv("SLPX_DENSE", "1")
    rng = np.random.default_rng(31)
    n, m_e = 48, 20
    for definite in (True, False):
        K, colptr, rowidx, vals = _kkt(rng, n, m_e, definite=definite, c22=1e-2 if definite else 0.0)
        scale = 1.0 + 0.1 * rng.random((batch, 1))
        ls = sa.System.linear_solver(n, m_e, colptr, rowidx, batch=batch)
        assert ls.info["ldlt_dense"] == 1
        ls.reset_regularization(1e-10)
        ls.set_matrix(vals[None, :] * scale)
        info, reg, nfact = ls.compute()
        assert (info == 0).all()
        if definite:
            assert nfact == 1 and np.all(reg == 0.0)
        else:
            assert np.all(reg[:, 0] > 0.0)
        rhs = rng.standard_normal((batch, n + m_e))
        ls.set_rhs(rhs)
        ls.solve()
        x = ls.get("p")
        for b in sorted({0, batch // 2, batch - 1}):
            _check_solution(K * scale[b, 0], n, reg[b], rhs[b], x[b])
        ls.close()
