#!/usr/bin/env python3
"""Back-to-back launches of a single problem's step (slpx_system_time_fused_step): tape sweep and
the factor + solve launch, for A/B runs on ONE box (SLPX_LIB=<other build>, SLPX_LDLT_MF=0, ...).

    PYTHONPATH=$PWD python profiles/mf_time.py [N ... | gfold]
"""
import sys

import numpy as np

import sleipnir_amd as sa
from tests.support import cases
from tests.support import models

for arg in sys.argv[1:] or ["1000"]:
    sa.lib().slpx_graph_reset()
    if arg == "gfold":
        from tests.support import gfold, model

        N = "g-fold N=100"
        pp = gfold.build(model.Model(model.ProductBackend("gpu")), 100).p
    else:
        N = int(arg)
        pp = models.cart_pole(N, 5.0 / N)
    sy = sa.System(pp, batch=1, device=0)
    n, me, mi = sy.info["n"], sy.info["m_e"], sy.info["m_i"]
    x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
    sy.set_state(x, s, y, z, np.array([mu]))
    ts = [sy.time_fused_step(200) for _ in range(5)]
    best = min(ts, key=lambda t: t["kkt_factor_solve"])
    print(N, "factor+solve launch us:", " ".join(f"{1e3 * t['kkt_factor_solve']:.2f}" for t in ts),
          "| sweep %.2f" % (1e3 * best["sweep"]), "one launch" if best["one_launch"] else "two launches",
          "multifrontal" if best["multifrontal"] else "pair lists")
    sy.close()
    pp.close()
