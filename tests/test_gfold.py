"""BASELINE config 5 (g-fold powered-descent OCP: quadratic cone-type inequality rows, so
the general AᵢᵀΣAᵢ product and dense Hessian blocks are exercised — SURVEY.md §8).
CPU tier: compiled plans vs oracle on a short horizon.  GPU tier: N = 100 (the benchmark
horizon of SURVEY.md §8d) Newton-step parity and the whole solve against the oracle's."""
import numpy as np
import pytest

from tests.support import gfold, model, parity


def build_both(N):
    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    return gfold.build(mo, N), gfold.build(mp, N)


def test_gfold_dimensions_and_types(fresh):
    po, pp = build_both(6)
    assert po.p.dims == gfold.dims(6) == pp.p.dims
    assert po.types() == pp.types() == (2, 2, 3)  # LINEAR cost, LINEAR equalities, QUADRATIC inequalities
    assert np.array_equal(po.p.get_x(), pp.p.get_x())
    lo, hi = gfold.n_range()
    assert lo <= 100 <= hi  # the benchmark horizon lies inside the search interval (main.cpp:397-398)


@pytest.mark.parametrize("case", ["step0", "interior"])
def test_gfold_plans_match_oracle(fresh, hostcheck, case):
    po, pp = build_both(8)
    hc = hostcheck.HostCheck(pp.p)
    try:
        parity.check_newton_step(hc, po.p, case)
    finally:
        hc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["step0", "interior"])
def test_gfold_n100_newton_step_gpu(fresh, slpx, case):
    po, pp = build_both(100)
    system = slpx.System(pp.p, batch=1, device=0)
    try:
        errs = parity.check_newton_step(parity.GpuBackend(system), po.p, case, verbose=True)
        assert errs["p"] <= 1e-6
    finally:
        system.close()


@pytest.mark.gpu
def test_gfold_n100_solve_gpu(fresh, slpx):
    po, pp = build_both(100)
    so = po.solve()
    sp = pp.solve()
    assert sp == so == model.NlpProblem.SUCCESS
    xo, xp = po.p.get_x(), pp.p.get_x()
    N = 100
    fuel_o, fuel_p = xo[-N:].sum(), xp[-N:].sum()  # cost = Σσ (main.cpp:380)
    assert abs(fuel_o - fuel_p) <= 1e-6 * max(1.0, abs(fuel_o))
    assert np.max(np.abs(xo - xp)) <= 1e-4 * max(1.0, float(np.max(np.abs(xo))))
