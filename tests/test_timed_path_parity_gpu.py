"""GPU tier, VERDICT r02 item 1: the kernels the bench TIMES, under the oracle, at BASELINE sizes.

`parity.check_newton_step` drives the stand-alone stages (sweep / assemble / rhs / factor / solve /
backsub); what `bench.py` measures is `slpx_newton_step`: for one problem the generated tape kernel
followed by `ldlt_mf_step_kernel` (KKT evaluated inside the factorization, backward solve and
back-substitution in the same launch), at N=5000 the two-launch variant, for batches of 64 and
more the batch-interleaved `ldlt_*_il_kernel`s.  Each of those paths here against
`oracle newton_step` (interior_point.hpp:426-482) — (delta, gamma), inertia, p, p_s, p_z:

  config 2  cart-pole N=1000, one problem  — one launch
  config 3  cart-pole N=5000               — whatever the library picks at that size
  config 5  g-fold N=100                   — one launch, general AiT Sigma Ai products
  batch     512 x cart-pole N=1000, items 0, 255, 511 — interleaved kernels
"""
import os

import numpy as np
import pytest

from tests.support import cases, gfold, model, parity

pytestmark = pytest.mark.gpu


def _step_and_check(system, op, case, label):
    n, me, mi = system.info["n"], system.info["m_e"], system.info["m_i"]
    scales = op.scaling()
    system.set_scaling(scales)
    state = cases.newton_state(case, op.get_x(), n, me, mi, scales[0])
    x, s, y, z, mu = state
    system.set_state(x, s, y, z, np.array([mu]))
    system.reset_regularization()
    assert np.all(system.newton_step(True) == 0)
    return parity.check_timed_step(system, op, state, verbose=True, label=label)


@pytest.mark.parametrize("case", ["step0", "interior"])
def test_config2_n1000_fused_step_against_oracle(fresh, slpx, orc, case):
    pp, op = cases.build_pair("cart_pole", 1000, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    try:
        # the path BENCH times at this size (unless profiles/switch_matrix.sh asked for the launches apart)
        apart = any(os.environ.get(k) == "0" for k in ("SLPX_LDLT_MF", "SLPX_FUSE_LAUNCHES", "SLPX_FUSE_KKT", "SLPX_FUSE_BACKSUB",
                                                        "SLPX_SINGLE_LAUNCH"))
        assert system.time_fused_step(1)["one_launch"] or apart
        errs = _step_and_check(system, op, case, f"N=1000 {case} (one launch)")
        assert errs["resid"] <= 1e-10
    finally:
        system.close()


def test_config3_n5000_timed_step_against_oracle(fresh, slpx, orc):
    pp, op = cases.build_pair("cart_pole", 5000, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    try:
        one = system.time_fused_step(1)["one_launch"]
        errs = _step_and_check(system, op, "interior", f"N=5000 interior ({'one launch' if one else 'two launches'})")
        assert errs["resid"] <= 1e-10
    finally:
        system.close()


@pytest.mark.parametrize("case", ["step0", "interior"])
def test_config5_gfold_n100_fused_step_against_oracle(fresh, slpx, case):
    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    po, pp = gfold.build(mo, 100), gfold.build(mp, 100)
    system = slpx.System(pp.p, batch=1, device=0)
    try:
        _step_and_check(system, po.p, case, f"g-fold N=100 {case}")
    finally:
        system.close()


@pytest.mark.parametrize("N,B,items", [(1000, 512, (0, 255, 511)), (500, 64, (0, 63)),
                                       (60, 200, (0, 63, 64, 191, 199))])
def test_batch_timed_step_items_against_oracle(fresh, slpx, orc, N, B, items):
    """512 x N=1000 (the batch in the bench line's `batched` object: interleaved LDLT kernels),
    config 4's per-GPU share, and the interleaved path's ragged case (200 = three chunks of 64 and
    one of 8): every checked item against its own oracle step."""
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    st = [cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + b) for b in range(B)]
    system = slpx.System(pp, batch=B, device=0)
    try:
        system.set_scaling(scales)
        system.set_state(*(np.stack([s[k] for s in st]) for k in range(4)), np.array([s[4] for s in st]))
        system.reset_regularization()
        assert np.all(system.newton_step(True) == 0)
        snap = parity.snapshot_step(system)
        for b in items:
            parity.check_timed_step(system, op, st[b], b=b, verbose=True, label=f"{B} x N={N} item {b}", snap=snap)
    finally:
        system.close()


# ---- the alternative paths, in the suite the driver runs -----------------
# The switches are read when a system is made (DESIGN.md §4 "Switches"): set for ONE system here,
# each against the oracle's whole step like the default path above — so that the kernels the
# default path does not reach (matrix-core update blocks, the pair-list one-launch step, the
# unchained launch order, the exact-structure fronts, 512-thread tasks) are under the oracle in
# GPUTEST, not only in the builder's profiles/switch_matrix.sh.

@pytest.mark.parametrize("env,expect", [
    ({"SLPX_LDLT_MF": "0"}, {"ldlt_multifrontal": 0}),                 # r02's pair-list step kernel
    ({"SLPX_CHAIN_TAPE": "0"}, {"ldlt_multifrontal": 1}),              # sweep and step kernel in one stream
    ({"SLPX_RELAX_ZEROS": "0"}, {"ldlt_multifrontal": 1}),             # exact structure of L: 20 levels of fronts
    ({"SLPX_MF_THREADS": "512"}, {"ldlt_multifrontal": 1}),            # the variant N=5000 runs
    ({"SLPX_FUSE_LAUNCHES": "0"}, {}),                                 # stand-alone assembly / factorization / solve
], ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) and any(k.startswith("SLPX") for k in v) else "")
def test_config2_alternative_paths_against_oracle(fresh, slpx, orc, monkeypatch, env, expect):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pp, op = cases.build_pair("cart_pole", 1000, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    try:
        for k, v in expect.items():
            assert cases.OUTER_SWITCHES or system.info[k] == v, (k, system.info[k])
        # several consecutive steps first: chained / unchained launch order is a property of a SEQUENCE
        n, me, mi = system.info["n"], system.info["m_e"], system.info["m_i"]
        scales = op.scaling()
        system.set_scaling(scales)
        x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
        system.set_state(x, s, y, z, np.array([mu]))
        assert np.all(system.newton_steps(3) == 0)
        errs = _step_and_check(system, op, "interior", f"N=1000 interior {env}")
        assert errs["resid"] <= 1e-10
    finally:
        system.close()


def test_config5_gfold_matrix_core_fronts_against_oracle(fresh, slpx, monkeypatch):
    """Row X1: every eligible front's update block through v_mfma_f64_16x16x4_f64
    (SLPX_MFMA_MIN_ENTRIES=0; the default threshold keeps only the large blocks there) — the plan must
    say that fronts are on the matrix cores, and the step must be the oracle's."""
    monkeypatch.setenv("SLPX_MFMA_MIN_ENTRIES", "0")
    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    po, pp = gfold.build(mo, 100), gfold.build(mp, 100)
    system = slpx.System(pp.p, batch=1, device=0)
    try:
        assert cases.OUTER_SWITCHES or (system.info["ldlt_multifrontal"] == 1 and system.info["ldlt_mfma_fronts"] >= 50), system.info
        for case in ("step0", "interior"):
            _step_and_check(system, po.p, case, f"g-fold N=100 {case}, all eligible fronts on the matrix cores "
                                                f"({system.info['ldlt_mfma_fronts']} of {system.info['ldlt_fronts']})")
    finally:
        system.close()


@pytest.mark.parametrize("kind,N", [("cart_pole", 1000), ("cart_pole", 100), ("cart_pole", 5000), ("gfold", 100)])
def test_a_new_right_hand_side_through_the_fronts(fresh, slpx, orc, monkeypatch, kind, N):
    """ldlt_mf_solve_kernel (r05; second-order corrections, interior_point.hpp:611-619, the multiplier estimate,
    slpx_system_solve): K p = b for a right-hand side that was not there when the step kernel factored K — against
    the residual of the regularized system, against the pair-list kernels on the same factor (SLPX_MF_SOLVE=0, a
    second system), and repeated (the hand-over slots and both x buffers are left armed by every launch)."""
    def build():
        if kind == "gfold":
            mo = model.Model(model.OracleBackend())
            mo.be.reset()
            mp = model.Model(model.ProductBackend("hostcheck"))
            mp.be.reset()
            return gfold.build(mp, N).p, gfold.build(mo, N).p
        return cases.build_pair(kind, N, slpx, orc)

    pp, op = build()
    system = slpx.System(pp, batch=1, device=0)
    monkeypatch.setenv("SLPX_MF_SOLVE", "0")
    other = slpx.System(pp, batch=1, device=0)
    monkeypatch.delenv("SLPX_MF_SOLVE")
    try:
        # (under an outer switch that takes the fronts away — the switch matrix — both systems solve on the pair lists:
        # the residual checks below still hold)
        assert cases.OUTER_SWITCHES or system.info["ldlt_multifrontal"] == 1
        n, me, mi = system.info["n"], system.info["m_e"], system.info["m_i"]
        scales = op.scaling()
        state = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
        x, s, y, z, mu = state
        results = []
        for sy in (system, other):
            sy.set_scaling(scales)
            sy.set_state(x, s, y, z, np.array([mu]))
            sy.reset_regularization()
            assert np.all(sy.newton_step(True) == 0)
        delta, gamma = (float(v) for v in system.regularization()[0])
        lhs = system.get("lhs")[0]
        lcp, lri = system.pattern(5)
        Kreg = cases.regularized(lcp, lri, lhs, n, delta, gamma)
        k_inf = float(np.max(cases.lower_csc_matvec(lcp, lri, np.abs(Kreg), np.ones(n + me))))
        rng = np.random.default_rng(11)
        for trial in range(3):
            b = rng.uniform(-1, 1, n + me)
            ps = []
            for sy in (system, other):
                sy.set_rhs(b[None, :])
                sy.solve()
                ps.append(sy.get("p")[0])
            p_fronts, p_pairs = ps
            resid = float(np.max(np.abs(cases.lower_csc_matvec(lcp, lri, Kreg, p_fronts) - b)))
            assert resid <= 1e-9 * max(1.0, k_inf * float(np.max(np.abs(p_fronts)))), (trial, resid)
            resid_pairs = float(np.max(np.abs(cases.lower_csc_matvec(lcp, lri, Kreg, p_pairs) - b)))
            assert resid <= max(1e-12 * k_inf * float(np.max(np.abs(p_fronts))), 10.0 * resid_pairs), (trial, resid, resid_pairs)
        # and the step after it is the step before it, to the bit: nothing the solves left behind disturbs a factorization
        p_before = system.get("p")[0].copy()
        system.reset_regularization()
        assert np.all(system.newton_step(True) == 0)
        first = system.get("p")[0].copy()
        system.set_rhs(rng.uniform(-1, 1, n + me)[None, :])
        system.solve()
        system.reset_regularization()
        assert np.all(system.newton_step(True) == 0)
        assert np.array_equal(system.get("p")[0], first)
        del p_before
    finally:
        system.close()
        other.close()
