#!/usr/bin/env python3
"""Whole solves over and over (twin attempts, look-ahead iteration): every repetition of a horizon must end with
the same status, iteration count and solution, to the bit, and none may hang or fail.
    PYTHONPATH=$PWD python profiles/solve_soak.py [seconds per horizon]"""
import hashlib
import sys
import time

import numpy as np

import sleipnir_amd as sa
from tests.support import models

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
for N in (50, 100, 150, 300, 500, 700):
    first = None
    reps = 0
    distinct = {}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        sa.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        st, rep = pp.solve()
        key = (st, int(rep["iterations"]), int(rep["factorizations"]), hashlib.sha256(np.ascontiguousarray(pp.get_x()).tobytes()).hexdigest())
        distinct[key] = distinct.get(key, 0) + 1
        if first is None:
            first = key
        reps += 1
        pp.close()
    print(f"N={N}: {reps} solves in {time.perf_counter() - t0:.1f} s, status {first[0]}, {first[1]} iterations, {first[2]} factorizations, "
          f"distinct results {len(distinct)}", flush=True)
    if len(distinct) > 1:
        for k, c in distinct.items():
            print(f"    {c} x status {k[0]}, {k[1]} iterations, {k[2]} factorizations, x {k[3][:16]}", flush=True)
