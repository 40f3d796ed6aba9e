"""GPU tier, VERDICT r02 item 1: the kernels the bench TIMES, under the oracle, at BASELINE sizes.

`parity.check_newton_step` drives the stand-alone stages (sweep / assemble / rhs / factor / solve /
backsub); what `bench.py` measures is `slpx_newton_step`: for one problem the generated tape kernel
followed by `ldlt_factor_solve_kernel` (KKT evaluated inside the factorization, backward solve and
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
        apart = any(os.environ.get(k) == "0" for k in ("SLPX_FUSE_LAUNCHES", "SLPX_FUSE_KKT", "SLPX_FUSE_BACKSUB",
                                                        "SLPX_FUSE_SOLVE", "SLPX_SINGLE_LAUNCH"))
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
        for b in items:
            parity.check_timed_step(system, op, st[b], b=b, verbose=True, label=f"{B} x N={N} item {b}")
    finally:
        system.close()
