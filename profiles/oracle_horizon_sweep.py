#!/usr/bin/env python3
"""The CPU restatement of the reference algorithm (oracle/) on the BASELINE horizons, from the benchmark's
initial guess and from that guess multiplied by (1 + 1e-13 u) — status, iterations, seconds per run.  CPU only:
    PYTHONPATH=$PWD python profiles/oracle_horizon_sweep.py [N ...] > profiles/rNN_oracle_horizon_sweep.txt"""
import sys
import time

import numpy as np

from tests.support import oracle

NAMES = {0: "SUCCESS", -2: "LOCALLY_INFEASIBLE", -4: "FACTORIZATION_FAILED", -9: "MAX_ITERATIONS_EXCEEDED", -8: "TIMEOUT"}
horizons = [int(a) for a in sys.argv[1:]] or [800, 900, 1000]
print("# oracle/ (CPU restatement of interior_point.hpp + feasibility_restoration.hpp), cart-pole swing-up, one core;")
print("# seed 0 = the benchmark's initial guess, seeds 1.. = that guess x (1 + 1e-13 u), u uniform in [-1, 1]")
for N in horizons:
    for k in range(3):
        oracle.lib().orc_reset()
        op = oracle.OracleProblem.cart_pole(N, 5.0 / N)
        x = op.get_x()
        if k:
            x = x * (1 + 1e-13 * np.random.default_rng(k).uniform(-1, 1, len(x)))
        op.set_x(x)
        t = time.time()
        st, stats = op.solve()
        print(f"N={N} seed {k}: status {st} {NAMES.get(int(st), '?')}, iterations {int(stats['iterations'])}, {time.time() - t:.1f} s", flush=True)
