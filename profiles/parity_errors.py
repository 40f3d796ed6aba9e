#!/usr/bin/env python3
"""The verbose record behind tests/test_timed_path_parity_gpu.py: one line per
configuration with what `parity.check_timed_step` measured on THIS build — kappa, the residuals, the
distance of the product's and of the oracle's step to the refined solution (p_vs_true / po_vs_true),
the distance after one refinement step with the device's own factors (p1_vs_true) and which branch of
the forward-error rule held.  On the GPU box:  python profiles/parity_errors.py > profiles/rNN_parity_errors.txt"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

import sleipnir_amd as slpx  # noqa: E402
from tests.support import cases, gfold, model, oracle as orc, parity  # noqa: E402

KEYS = ("delta", "gamma", "kappa", "resid", "resid_oracle", "p", "p_s", "p_z", "p_vs_true", "po_vs_true", "p1_vs_true", "p_vs_true_kappa_eps",
        "forward_rule", "D_rel_median", "D_rel_p90", "D_rel", "lhs", "rhs")


def line(label, errs):
    ratio = errs["p_vs_true"] / max(errs["po_vs_true"], 1e-300)
    body = "  ".join(f"{k}={errs[k]:.2e}" if isinstance(errs[k], float) else f"{k}={errs[k]}" for k in KEYS)
    print(f"{label:<58s} ratio={ratio:6.2f}  {body}", flush=True)


def fresh():
    orc.lib().orc_reset()
    slpx.lib().slpx_graph_reset()


def single(label, system, op, case, b=0):
    n, me, mi = system.info["n"], system.info["m_e"], system.info["m_i"]
    scales = op.scaling()
    system.set_scaling(scales)
    state = cases.newton_state(case, op.get_x(), n, me, mi, scales[0])
    x, s, y, z, mu = state
    system.set_state(x, s, y, z, np.array([mu]))
    system.reset_regularization()
    assert np.all(system.newton_step(True) == 0)
    line(label, parity.check_timed_step(system, op, state, label=label))


def cart_pole(N, case, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    fresh()
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    for k in env:
        del os.environ[k]
    tag = " ".join(f"{k}={v}" for k, v in env.items())
    single(f"cart-pole N={N} {case} {tag}".strip(), system, op, case)
    system.close()


def gfold_case(case, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    mo = model.Model(model.OracleBackend())
    mo.be.reset()
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    po, pp = gfold.build(mo, 100), gfold.build(mp, 100)
    system = slpx.System(pp.p, batch=1, device=0)
    for k in env:
        del os.environ[k]
    tag = " ".join(f"{k}={v}" for k, v in env.items())
    single(f"g-fold N=100 {case} {tag} (matrix-core fronts {system.info['ldlt_mfma_fronts']})".strip(), system, po.p, case)
    system.close()


def batch(N, B, items):
    fresh()
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    st = [cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + b) for b in range(B)]
    system = slpx.System(pp, batch=B, device=0)
    system.set_scaling(scales)
    system.set_state(*(np.stack([s[k] for s in st]) for k in range(4)), np.array([s[4] for s in st]))
    system.reset_regularization()
    assert np.all(system.newton_step(True) == 0)
    snap = parity.snapshot_step(system)
    for b in items:
        label = f"{B} x cart-pole N={N} item {b}"
        line(label, parity.check_timed_step(system, op, st[b], b=b, label=label, snap=snap))
    system.close()


if __name__ == "__main__":
    print("# check_timed_step on the kernels bench.py times; ratio = p_vs_true / po_vs_true; rule: direct = within 10 x the")
    print(f"# oracle's distance, refined = within {parity.FORWARD_ENVELOPE:.0f} kappa eps of the refined solution AND one device refinement step reaches the oracle's distance")
    for case in ("step0", "interior"):
        cart_pole(1000, case)
    for env in ({"SLPX_LDLT_MF": "0"}, {"SLPX_CHAIN_TAPE": "0"}, {"SLPX_RELAX_ZEROS": "0"}, {"SLPX_MF_THREADS": "512"}):
        cart_pole(1000, "interior", env)
    cart_pole(5000, "interior")
    for case in ("step0", "interior"):
        gfold_case(case)
    gfold_case("interior", {"SLPX_MFMA_MIN_ENTRIES": "0"})
    batch(1000, 512, (0, 255, 511))
    batch(500, 64, (0, 63))
