"""GPU tier, VERDICT r01 "parity gaps": what the first round computed but never asserted.

  * the product's inertia-correcting policy loop (slpx_ldlt_compute) against the oracle's own loop
    (sparse_regularized_ldlt.hpp:64-152) on the same matrix and the same elimination order: the
    (delta, gamma) sequence must end in the same place after the same number of factorizations —
    cart-pole N=100 / N=1000 and the indefinite fixture state, whose loop escalates;
  * config 3 (N=5000): a full oracle Newton step, every quantity including p, p_s, p_z;
  * config 4: batch items against the oracle, one by one;
  * the supernode settings of the LDLT (column levels; chains from 2 columns up) through the C-ABI.
(`parity.check_newton_step` itself now asserts p, p_s, p_z and the pivots, see its docstring.)
"""
import os
from pathlib import Path

import numpy as np
import pytest

from tests.support import cases, parity

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def _assembled(system, op, x, s, y, z, mu):
    scales = op.scaling()
    system.set_scaling(scales)
    system.set_state(x, s, y, z, np.array([mu]))
    system.sweep(True)
    system.assemble()
    system.rhs()
    return system.pattern(5), system.get("lhs")[0]


@pytest.mark.parametrize("N,case", [(100, "step0"), (100, "interior"), (1000, "step0"), (1000, "interior")])
def test_policy_loop_decisions_match_oracle(fresh, slpx, orc, N, case):
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    system = slpx.System(pp, batch=1, device=0)
    try:
        x, s, y, z, mu = cases.newton_state(case, op.get_x(), n, me, mi, op.scaling()[0])
        (cp, ri), lhs = _assembled(system, op, x, s, y, z, mu)
        d, g, nf, onf = parity.check_policy_loop(system, (cp, ri, lhs), n, me)
        print(f"N={N} {case}: (delta, gamma) = ({d:g}, {g:g}), factorizations product {nf} / oracle {onf}")
        assert d > 0.0  # these states need the Hessian regularization
    finally:
        system.close()


def test_policy_loop_escalates_like_the_oracle_on_the_indefinite_fixture(fresh, slpx, orc):
    """cart_pole_N8_indefinite.npz: a state whose reduced Hessian is indefinite, so the loop has
    to raise delta several times (sparse_regularized_ldlt.hpp:124-130); also with gamma_min = 0,
    the restoration setting (interior_point.hpp:352)."""
    fx = dict(np.load(GOLDEN / "cart_pole_N8_indefinite.npz"))
    N = int(fx["N"])
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    system = slpx.System(pp, batch=1, device=0)
    try:
        (cp, ri), lhs = _assembled(system, op, fx["x"], fx["s"], fx["y"], fx["z"], float(fx["mu"]))
        d, g, nf, onf = parity.check_policy_loop(system, (cp, ri, lhs), n, me)
        assert d >= fx["chosen"][0] and nf >= 2, (d, nf)
        d0, g0, _, _ = parity.check_policy_loop(system, (cp, ri, lhs), n, me, gamma_min=0.0)
        print(f"indefinite fixture: gamma_min 1e-10 -> ({d:g}, {g:g}) after {nf}; gamma_min 0 -> ({d0:g}, {g0:g})")
    finally:
        system.close()


def test_config3_n5000_full_oracle_step(fresh, slpx, orc):
    """config 3 ran no oracle factorization.  The whole step against the oracle
    (same permutation): AD values, lhs, rhs, pivots, p, p_s, p_z."""
    pp, op = cases.build_pair("cart_pole", 5000, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    try:
        errs = parity.check_newton_step(parity.GpuBackend(system), op, "interior", verbose=True)
        assert errs["resid"] <= 1e-10
    finally:
        system.close()


def test_config4_batch_items_match_oracle(fresh, slpx, orc):
    """Config 4's per-GPU share (64 x cart-pole N=500): items 0, 17, 40 and 63 against the
    oracle run on that item's state with the product's permutation — (delta, gamma), lhs, rhs, p,
    p_s, p_z."""
    N, B = 500, 64
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    st = [cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + b) for b in range(B)]
    sysb = slpx.System(pp, batch=B, device=0)
    try:
        sysb.set_scaling(scales)
        sysb.set_state(*(np.stack([s_[k] for s_ in st]) for k in range(4)), np.array([s_[4] for s_ in st]))
        sysb.reset_regularization()
        assert np.all(sysb.newton_step(True) == 0)
        reg = sysb.regularization()
        out = {k: sysb.get(k) for k in ("p", "p_s", "p_z", "lhs", "rhs")}
        cp, ri = sysb.pattern(5)
        perm = sysb.perm()
        for b in (0, 17, 40, 63):
            x, s, y, z, mu = st[b]
            info, _ = op.newton_step(x, s, y, z, mu, True, perm)
            assert info == 0
            d, g, _, _ = op.reg()
            assert (d, g) == (reg[b, 0], reg[b, 1])
            assert cases.max_rel(out["rhs"][b], op.vec("rhs")) <= 1e-10
            Kreg = cases.regularized(cp, ri, out["lhs"][b], n, d, g)
            kappa = cases.cond_inf_estimate(cp, ri, Kreg)
            po = op.vec("p")
            eta = [float(np.max(np.abs(cases.lower_csc_matvec(cp, ri, Kreg, v) - out["rhs"][b]))) for v in (out["p"][b], po)]
            k_inf = float(np.max(cases.lower_csc_matvec(cp, ri, np.abs(Kreg), np.ones(n + me))))
            scale = max(1.0, float(np.max(np.abs(out["rhs"][b]))), k_inf * float(np.max(np.abs(po))))
            tol = max(1e-8, 2.0 * kappa * (eta[0] + eta[1]) / scale)
            assert eta[0] / scale <= 1e-10
            assert cases.max_rel(out["p"][b], po) <= tol, (b, cases.max_rel(out["p"][b], po), tol)
            assert cases.max_rel(out["p_s"][b], op.vec("p_s")) <= 100 * tol
            assert cases.max_rel(out["p_z"][b], op.vec("p_z")) <= 1e4 * tol
    finally:
        sysb.close()


@pytest.mark.parametrize("env", [{"SLPX_SN_MIN_WIDTH": "2"}, {"SLPX_SN_MIN_WIDTH": "8"}])
@pytest.mark.parametrize("kind,N", [("cart_pole", 100), ("gfold", 30)])
def test_supernode_settings(fresh, slpx, orc, monkeypatch, env, kind, N):
    """Chains from 2 columns up, only the widest chains: the same Newton step
    (factorization, the solve that rides in it, the re-solve with a new right-hand side)."""
    from tests.support import gfold, model

    monkeypatch.delenv("SLPX_SN_MIN_WIDTH", raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if kind == "gfold":
        mo = model.Model(model.OracleBackend())
        mo.be.reset()
        mp = model.Model(model.ProductBackend("hostcheck"))
        mp.be.reset()
        op, pp = gfold.build(mo, N).p, gfold.build(mp, N).p
    else:
        pp, op = cases.build_pair(kind, N, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    try:
        if not cases.OUTER_SWITCHES:  # (SLPX_RELAX_ZEROS=0, SLPX_SN_MAX_WIDTH=6 leave no chain of 8 columns)
            assert system.info["ldlt_widest_supernode"] >= 2
        parity.check_newton_step(parity.GpuBackend(system), op, "interior", verbose=True)
        # newton_step(): the rhs rides in the factorization; solve(): forward + backward kernels
        system.reset_regularization()
        assert system.newton_step(True)[0] == 0
        p_fused = system.get("p")[0].copy()
        system.solve()
        p_full = system.get("p")[0]
        assert cases.max_rel(p_fused, p_full) <= 1e-9
    finally:
        system.close()


@pytest.mark.parametrize("mf", ["1", "0"])
@pytest.mark.parametrize("kind,N", [("cart_pole", 37), ("cart_pole", 300), ("flywheel", 50)])
def test_system_evaluated_inside_the_factorization_equals_the_assembled_one(fresh, slpx, orc, monkeypatch, kind, N, mf):
    """r02: a single problem's Newton step no longer assembles lhs / rhs in memory — the
    factorization's tasks evaluate the entries they own from the AD sweep's V (device.hpp:
    KktFuse; kkt_kernels.h).  SLPX_FUSE_KKT_STORE=1 makes them also write what they evaluated
    where the assembly kernels would have: the two must agree to the bit (and both are what the
    oracle comparison of tests/support/parity.py sees).  The steps: to the bit with the pair-list
    kernels on both sides (SLPX_LDLT_MF=0); the multifrontal step (r03, the default) sums the
    updates front by front instead of pair by pair: same inertia, pivots and step to rounding."""
    pp, op = cases.build_pair(kind, N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    got = {}
    monkeypatch.setenv("SLPX_LDLT_MF", mf)
    for mode, env in (("inline", {"SLPX_FUSE_KKT_STORE": "1"}), ("assembled", {"SLPX_FUSE_KKT": "0"})):
        for k in ("SLPX_FUSE_KKT_STORE", "SLPX_FUSE_KKT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        system = slpx.System(pp, batch=1, device=0)
        system.set_scaling(scales)
        system.set_state(x, s, y, z, np.array([mu]))
        assert system.newton_step(True)[0] == 0
        got[mode] = {k: system.get(k)[0].copy() for k in ("lhs", "rhs", "p", "p_s", "p_z", "D")}
        got[mode]["mf"] = system.time_fused_step(1)["multifrontal"]
        system.close()
    # (under a switch of profiles/switch_matrix.sh that takes the one-launch step away — SLPX_LDLT_MF=0,
    # SLPX_FUSE_*=0 — there is no multifrontal step to ask for)
    is_mf = bool(got["inline"]["mf"])
    switched = any(os.environ.get(k) == "0" for k in ("SLPX_LDLT_MF", "SLPX_FUSE_LAUNCHES", "SLPX_FUSE_BACKSUB", "SLPX_SINGLE_LAUNCH"))
    assert (is_mf == (mf == "1") or switched) and not got["assembled"]["mf"] and not (is_mf and mf == "0")
    for k in ("lhs", "rhs"):
        assert np.array_equal(got["inline"][k], got["assembled"][k]), k
    if not is_mf:
        for k in ("D", "p"):
            assert np.array_equal(got["inline"][k], got["assembled"][k]), k
        # (p_s, p_z: the back-substitution riding in the solve's launch sums A_i p in the same order)
        for k in ("p_s", "p_z"):
            assert cases.max_rel(got["inline"][k], got["assembled"][k]) <= 1e-14, k
    else:
        Di, Da = got["inline"]["D"], got["assembled"]["D"]
        assert np.array_equal(np.sign(Di), np.sign(Da))
        assert np.median(np.abs(Di - Da) / np.abs(Da)) <= 1e-13
        for k in ("p", "p_s", "p_z"):
            assert cases.max_rel(got["inline"][k], got["assembled"][k]) <= 1e-6, k
