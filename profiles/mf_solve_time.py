#!/usr/bin/env python3
"""A new right-hand side (second-order corrections, the multiplier estimate, slpx_ldlt_solve): through the fronts
(ldlt_mf_solve_kernel, one launch; r05) against the pair lists (ldlt_fwd_kernel + ldlt_bwd_kernel on L in memory;
SLPX_MF_SOLVE=0), same box, same factor.  Wall time of K back-to-back solves, host launches included.
    PYTHONPATH=$PWD python profiles/mf_solve_time.py [N ...]"""
import os
import sys
import time

import numpy as np

import sleipnir_amd as sa
from tests.support import cases, models

K = 300
for arg in sys.argv[1:] or ["500"]:
    N = int(arg)
    row = []
    for env in ("1", "0"):
        os.environ["SLPX_MF_SOLVE"] = env
        sa.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        sy = sa.System(pp, batch=1, device=0)
        n, me, mi = sy.info["n"], sy.info["m_e"], sy.info["m_i"]
        x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
        sy.set_state(x, s, y, z, np.array([mu]))
        sy.reset_regularization()
        assert np.all(sy.newton_step(True) == 0)
        sy.set_rhs(np.random.default_rng(3).uniform(-1, 1, (1, n + me)))
        for _ in range(20):
            sy.solve()
        sy.get("p")
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(K):
                sy.solve()
            sy.get("p")
            best = min(best, (time.perf_counter() - t0) / K)
        row.append(best)
        sy.close()
        pp.close()
    print(f"cart-pole N={N}: solve of a new right-hand side {1e6 * row[0]:.1f} us through the fronts, {1e6 * row[1]:.1f} us on the pair lists", flush=True)
