#!/usr/bin/env python3
"""Where do the fronts lose the decimal?  (at the seeded interior state of cart-pole N=1000 the
multifrontal step lands 19 x further from the refined solution than the oracle, the pair lists 7 x.)
The same measurement over MANY seeded interior states instead of one: for each state the distance of the
oracle's step, of the product's step (host interpreter of the compiled plan — the arithmetic of the kernels, front by
front — or, with `gpu`, the kernels themselves) and, for both, the distance relative to kappa * eps.  All three codes
factor the SAME matrix in the SAME elimination order (the product's permutation is handed to the oracle); they differ
in the order of the sums.
    SLPX_LDLT_MF=1 python profiles/forward_error_sweep.py host 1000 24      # fronts, host interpreter
    SLPX_LDLT_MF=0 python profiles/forward_error_sweep.py host 1000 24      # column/pair-list plan
    python profiles/forward_error_sweep.py gpu 1000 24                        # the step kernel itself (GPU box)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

import sleipnir_amd as slpx  # noqa: E402
from tests.support import cases, hostcheck, oracle as orc, parity  # noqa: E402


def main(backend_kind="host", N=1000, n_seeds=24):
    orc.lib().orc_reset()
    slpx.lib().slpx_graph_reset()
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    if backend_kind == "gpu":
        system = slpx.System(pp, batch=1, device=0)
        be = parity.GpuBackend(system)
    else:
        be = hostcheck.HostCheck(pp)
    scales = op.scaling()
    be.set_scaling(scales)
    perm = be.perm()
    lcp, lri = be.pattern(5)
    eps = np.finfo(float).eps
    rows = []
    print(f"# cart-pole N={N}, {n_seeds} seeded interior states, backend {backend_kind} "
          f"(multifrontal plan: {int(be.info.get('ldlt_multifrontal', -1)) if hasattr(be, 'info') else '?'})")
    print("# seed  kappa      oracle/true  product/true  ratio   oracle/(kappa eps)  product/(kappa eps)  resid_o   resid_p")
    for seed in range(n_seeds):
        x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + seed)
        info, _ = op.newton_step(x, s, y, z, mu, True, perm)
        assert info == 0
        delta, gamma, _, _ = op.reg()
        be.sweep(x, y, z, True)
        lhs = be.assemble(s, z)
        rhs = be.rhs(s, y, z, mu)
        be.factor(delta, gamma)
        p = be.solve()
        po = op.vec("p")
        Kreg = cases.regularized(lcp, lri, lhs, n, delta, gamma)
        p_true = cases.refined_solution(lcp, lri, Kreg, rhs)
        kappa = cases.cond_inf_estimate(lcp, lri, Kreg)
        k_inf = float(np.max(cases.lower_csc_matvec(lcp, lri, np.abs(Kreg), np.ones_like(rhs))))
        scale = max(1.0, float(np.max(np.abs(rhs))), k_inf * float(np.max(np.abs(po))))
        ro = float(np.max(np.abs(cases.lower_csc_matvec(lcp, lri, Kreg, po) - rhs))) / scale
        rp = float(np.max(np.abs(cases.lower_csc_matvec(lcp, lri, Kreg, p) - rhs))) / scale
        do, dp = cases.max_rel(po, p_true), cases.max_rel(p, p_true)
        rows.append((kappa, do, dp))
        print(f"  {seed:3d}  {kappa:9.2e}  {do:11.2e}  {dp:12.2e}  {dp / do:6.2f}  {do / (kappa * eps):18.3f}  {dp / (kappa * eps):19.3f}  {ro:8.1e}  {rp:8.1e}",
              flush=True)
    r = np.array(rows)
    ratio = r[:, 2] / r[:, 1]
    print(f"# ratio product/oracle: min {ratio.min():.2f}  median {np.median(ratio):.2f}  geometric mean {np.exp(np.mean(np.log(ratio))):.2f}  max {ratio.max():.2f}")
    print(f"# distance / (kappa eps): oracle median {np.median(r[:, 1] / (r[:, 0] * eps)):.3f} max {np.max(r[:, 1] / (r[:, 0] * eps)):.3f};"
          f" product median {np.median(r[:, 2] / (r[:, 0] * eps)):.3f} max {np.max(r[:, 2] / (r[:, 0] * eps)):.3f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "host", int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
         int(sys.argv[3]) if len(sys.argv) > 3 else 24)
