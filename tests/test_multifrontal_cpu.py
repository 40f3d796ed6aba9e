"""CPU tier: the multifrontal plan (csrc/ldlt_symbolic.hpp: LdltFront — dense fronts, update blocks
handed from child to parent front, relaxed supernodes, table-driven LDS addressing) interpreted
on the host exactly as csrc/ldlt_mf_kernels.h runs it (tests/support/hostcheck.cpp: hc_factor_mf,
hc_backward_mf: every table word a byte offset into the task's first 64 KB of LDS) must reproduce
the oracle's Newton step (interior_point.hpp:426-482, sparse_regularized_ldlt.hpp:64-161) — the
same check the pair-list plans get in test_plans_cpu.py."""
from pathlib import Path

import numpy as np
import pytest

from tests.support import cases, parity

GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.fixture()
def mf(monkeypatch):
    monkeypatch.setenv("SLPX_LDLT_MF", "1")


@pytest.mark.parametrize("kind,N", [("cart_pole", 4), ("cart_pole", 37), ("cart_pole", 120), ("flywheel", 50)])
@pytest.mark.parametrize("case", ["step0", "interior"])
def test_multifrontal_plan_matches_oracle(fresh, slpx, orc, hostcheck, mf, kind, N, case):
    pp, op = cases.build_pair(kind, N, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    plan = hc.mf_plan()
    assert plan["built"] and plan["fronts"] > 0 and plan["most_rows"] <= 64 and plan["widest"] <= 8
    parity.check_newton_step(hc, op, case)


@pytest.mark.parametrize("zeros", ["0", "4", "32"])
def test_relaxed_supernodes_trade_explicit_zeros_for_levels(fresh, slpx, orc, hostcheck, mf, monkeypatch, zeros):
    """LdltOptions::relax_zeros: a supernode joins the one its parent column heads at the price of
    explicit zeros — fewer, wider fronts and fewer levels on the critical path, the same step."""
    monkeypatch.setenv("SLPX_RELAX_ZEROS", zeros)
    pp, op = cases.build_pair("cart_pole", 60, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    parity.check_newton_step(hc, op, "interior")
    plan, levels = hc.mf_plan(), hc.supernode_plan()["critical_levels"]
    monkeypatch.setenv("SLPX_RELAX_ZEROS", "0")
    slpx.lib().slpx_graph_reset()
    orc.lib().orc_reset()
    pq, _ = cases.build_pair("cart_pole", 60, slpx, orc)
    exact = hostcheck.HostCheck(pq)
    plan0, levels0 = exact.mf_plan(), exact.supernode_plan()["critical_levels"]
    if zeros == "0":
        assert plan["nnz_L"] == plan0["nnz_L"] and plan["fronts"] == plan0["fronts"]
    else:
        assert plan["nnz_L"] > plan0["nnz_L"] and plan["fronts"] < plan0["fronts"] and levels <= levels0


def test_multifrontal_backward_solve_runs_on_what_the_factorization_left(fresh, slpx, orc, hostcheck, mf):
    """The fused step's solve reads U and 1/d in place (hc_backward_mf) — it must agree with the full
    forward + backward substitution from L in memory (the pair-list solve kernels)."""
    pp, op = cases.build_pair("cart_pole", 37, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    n, me, mi = hc.n, hc.m_e, hc.m_i
    scales = op.scaling()
    hc.set_scaling(scales)
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    hc.sweep(x, y, z, True)
    hc.assemble(s, z)
    hc.rhs(s, y, z, mu)
    hc.factor(1e-4, 1e-10)
    p_full = hc.solve()
    p_fused = hc.solve_after_factor()
    assert cases.max_rel(p_fused, p_full) <= 1e-9
    hc.close()


def test_multifrontal_plan_on_the_indefinite_fixture(fresh, slpx, orc, hostcheck, mf):
    """cart_pole_N8_indefinite.npz: the inertia along the (delta, gamma) ladder from the fronts equals
    the fixture's (numpy eigvalsh on the dense KKT matrix, tests/golden/make_fixtures.py)."""
    fx = dict(np.load(GOLDEN / "cart_pole_N8_indefinite.npz"))
    pp, op = cases.build_pair("cart_pole", int(fx["N"]), slpx, orc)
    hc = hostcheck.HostCheck(pp)
    n, me = hc.n, hc.m_e
    scales = op.scaling()
    hc.set_scaling(scales)
    hc.sweep(fx["x"], fx["y"], fx["z"], True)
    hc.assemble(fx["s"], fx["z"])
    hc.rhs(fx["s"], fx["y"], fx["z"], float(fx["mu"]))
    for (delta, gamma), inertia in zip(fx["ladder"], fx["ladder_inertia"]):
        _, stats = hc.factor(float(delta), float(gamma))
        if delta == 0.0 and stats[2] + stats[3] > 0:
            continue  # (an exactly zero pivot at delta = gamma = 0: the reference's NumericalIssue branch)
        assert tuple(int(v) for v in stats[:3]) == tuple(int(v) for v in inertia), (delta, gamma, stats, inertia)
    hc.close()


def test_gfold_fronts_reach_the_matrix_cores(fresh, hostcheck, mf, monkeypatch):
    """BASELINE config 5: the g-fold problem's separator chains give fronts of four and more pivot
    columns over hundreds of update entries — the ones ldlt_mf_kernels.h: mf_update_mfma takes as
    v_mfma_f64_16x16x4_f64 tiles (from LdltOptions::mfma_min_entries up); cart-pole has none."""
    from tests.support import gfold, model

    monkeypatch.setenv("SLPX_MFMA_MIN_ENTRIES", "128")
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    g = gfold.build(mp, 40)
    plan = hostcheck.HostCheck(g.p).mf_plan()
    assert plan["built"] and plan["mfma_fronts"] > 0 and plan["widest"] >= 4


@pytest.mark.parametrize("env", [{"SLPX_SN_DEEPEST": "0"}, {"SLPX_SN_DEEPEST": "1"}, {"SLPX_SN_DEEPEST": "2"},
                                 {"SLPX_SN_BALANCE": "0"}, {"SLPX_SN_MAX_WIDTH": "5"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_supernode_chain_rules_keep_the_step(fresh, slpx, orc, hostcheck, mf, monkeypatch, env):
    """r04's rules for which columns form a supernode (LdltOptions::chain_from_deepest_child: a column joins its
    parent's chain only if no sibling subtree is as deep as its own; balance_supernode_cuts: over-long chains in
    equal pieces; max_supernode_width) change the fronts, never the step: every variant against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pp, op = cases.build_pair("cart_pole", 120, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    plan = hc.mf_plan()
    assert plan["built"] and plan["widest"] <= int(env.get("SLPX_SN_MAX_WIDTH", 8))
    parity.check_newton_step(hc, op, "interior")


def test_chains_from_the_deepest_child_shorten_the_longest_pivot_chain(fresh, slpx, orc, hostcheck, mf, monkeypatch):
    """What the rule is for: a separator merged into a parent whose other child subtree is as deep puts its pivots
    behind that sibling.  With the rule the fronts are more and narrower (a 9-column chain at the end of the
    dissected stages is no longer one 8 + 1); levels may differ by one either way — what shrinks is the longest run
    of pivots, which a level count does not see (DESIGN.md section 4a has the kernel times)."""
    def plan_with(deepest):
        monkeypatch.setenv("SLPX_SN_DEEPEST", deepest)
        slpx.lib().slpx_graph_reset()
        orc.lib().orc_reset()
        pp, _ = cases.build_pair("cart_pole", 500, slpx, orc)
        hc = hostcheck.HostCheck(pp)
        return hc.mf_plan(), hc.supernode_plan()

    off, sn_off = plan_with("0")
    on, sn_on = plan_with("1")
    assert on["fronts"] > off["fronts"]
    assert abs(sn_on["critical_levels"] - sn_off["critical_levels"]) <= 1
    wide = lambda sn: sum(c for w, c in sn["width_hist"].items() if w >= 7)
    assert wide(sn_on) < wide(sn_off)


@pytest.mark.parametrize("kind,N", [("cart_pole", 37), ("cart_pole", 300), ("flywheel", 50)])
def test_a_new_right_hand_side_through_the_fronts(fresh, slpx, orc, hostcheck, mf, monkeypatch, kind, N):
    """ldlt_mf_solve_kernel as the host interprets it (hc_forward_mf + hc_backward_mf): the factor in memory back in
    the fronts' layout, the forward substitution as the right-hand-side row of the factorization, the step kernel's
    backward solve — against the solve that rode in the factorization, against the pair lists (SLPX_MF_SOLVE=0) and,
    for a right-hand side that was NOT there at the factorization, against a dense solve of the same matrix."""
    pp, op = cases.build_pair(kind, N, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    n, me, mi = hc.n, hc.m_e, hc.m_i
    scales = op.scaling()
    hc.set_scaling(scales)
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    hc.sweep(x, y, z, True)
    lhs = hc.assemble(s, z)
    rhs = hc.rhs(s, y, z, mu)
    hc.factor(1e-4, 1e-10)
    p_fused = hc.solve_after_factor()
    p_fronts = hc.solve()
    assert cases.max_rel(p_fronts, p_fused) <= 1e-9
    rng = np.random.default_rng(5)
    b2 = rng.uniform(-1, 1, n + me)
    hc.set_rhs(b2)
    p2 = hc.solve()
    monkeypatch.setenv("SLPX_MF_SOLVE", "0")
    p2_pairs = hc.solve()
    monkeypatch.delenv("SLPX_MF_SOLVE")
    assert cases.max_rel(p2, p2_pairs) <= 1e-9
    lcp, lri = hc.pattern(5)
    Kreg = cases.regularized(lcp, lri, lhs, n, 1e-4, 1e-10)
    resid = cases.lower_csc_matvec(lcp, lri, Kreg, p2) - b2
    assert np.max(np.abs(resid)) <= 1e-9 * max(1.0, float(np.max(np.abs(p2))) * float(np.max(np.abs(Kreg))))
    hc.close()


def test_the_baseline_models_get_the_multifrontal_plan(fresh, slpx, orc, hostcheck, mf):
    """A refused multifrontal plan (ldlt_symbolic.cpp: MfRefused, or a limit of the fronts' 16-bit addressing) degrades
    to the pair-list plan: a slower step, and nothing else would tell.  The BASELINE configurations must not get there
    (cart-pole N=1000 and N=5000, g-fold N=100)."""
    from tests.support import gfold, model

    for N in (1000, 5000):
        slpx.lib().slpx_graph_reset()
        orc.lib().orc_reset()
        pp, _ = cases.build_pair("cart_pole", N, slpx, orc)
        hc = hostcheck.HostCheck(pp)
        plan = hc.mf_plan()
        assert plan["built"] and plan["fronts"] > N, (N, plan)
        hc.close()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    hc = hostcheck.HostCheck(gfold.build(mp, 100).p)
    assert hc.mf_plan()["built"]
    hc.close()
