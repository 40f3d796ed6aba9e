#!/usr/bin/env python3
"""The same experiment as oracle_sensitivity.py on the PRODUCT (Problem::solve() on the GPU): the
benchmark's initial guess multiplied by (1 + 1e-13 u), six seeds per horizon (seed 0 =
unperturbed).  PYTHONPATH=$PWD python profiles/product_sensitivity.py [N ...]"""
import sys

import numpy as np

import sleipnir_amd as sa
from tests.support import models

for N in [int(a) for a in sys.argv[1:]] or [500, 1000]:
    for k in range(6):
        sa.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        x = pp.get_x()
        rng = np.random.default_rng(k)
        if k:
            x = x * (1 + 1e-13 * rng.uniform(-1, 1, len(x)))
        pp.set_x(x)
        st, rep = pp.solve()
        print(N, k, "status", st, "iterations", rep["iterations"], "restorations", rep["restorations"], flush=True)
        pp.close()
