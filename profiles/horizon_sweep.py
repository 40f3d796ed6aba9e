#!/usr/bin/env python3
"""Problem::solve() of the cart-pole swing-up over the BASELINE horizons: exit status, iterations,
wall time.  PYTHONPATH=$PWD python profiles/horizon_sweep.py [N ...]"""
import sys
import time

import sleipnir_amd as sa
from tests.support import models

Ns = [int(a) for a in sys.argv[1:]] or [50, 100, 150, 200, 300, 400, 500, 600, 700, 800, 900, 1000]
for N in Ns:
    sa.lib().slpx_graph_reset()
    pp = models.cart_pole(N, 5.0 / N)
    t0 = time.perf_counter()
    st, rep = pp.solve()
    print(f"N {N:5d} status {st:3d} iterations {rep['iterations']:5d} restorations {rep['restorations']:3d} "
          f"t_total {rep['t_total']:.3f} s (wall {time.perf_counter() - t0:.3f}) error {rep['final_error']:.2e} | "
          f"restoration: {rep['restoration_iterations']} iterations, setup {rep['t_restoration_setup']:.3f} s, "
          f"steps {rep['t_restoration']:.3f} s", flush=True)
    pp.close()
