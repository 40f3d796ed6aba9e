#!/usr/bin/env python3
"""Soak of the hand-over paths: many Newton steps from one state, every one of which must report
Success and give the same bits (any lost ordering between workgroups shows as a rare different
step).  Single problems take the one-launch (N=1000) and the two-launch (N=5000) path, the batch
the interleaved kernels.
    PYTHONPATH=$PWD python profiles/soak.py [steps at N=1000] [steps at N=5000] [steps of 512 x N=500]"""
import collections
import sys
import time

import numpy as np

import sleipnir_amd as sa
from tests.support import cases
from tests.support import models


def soak(N, B, steps):
    sa.lib().slpx_graph_reset()
    pp = models.cart_pole(N, 5.0 / N)
    n, me, mi = pp.dims
    st = [cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0, seed=cases.SEED + b) for b in range(B)]
    sy = sa.System(pp, batch=B, device=0)
    sy.set_state(*(np.stack([s[k] for s in st]) for k in range(4)), np.array([s[4] for s in st]))
    seen = collections.Counter()
    bad = 0
    t0 = time.time()
    chunk = 200 if B == 1 else 20
    for _ in range(0, steps, chunk):
        info = sy.newton_steps(chunk, True, True)
        bad += int(np.count_nonzero(info))
        seen[sy.get("p").tobytes()] += 1
    dt = time.time() - t0
    print(f"{B} x N={N}: {steps} steps in {dt:.1f} s ({B * steps / dt:.0f} steps/s incl. the read-backs), "
          f"failed {bad}, distinct results {len(seen)}", flush=True)
    sy.close()
    pp.close()
    return bad == 0 and len(seen) == 1


a = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
c = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
ok = soak(1000, 1, a) and soak(5000, 1, b) and soak(500, 512, c)
sys.exit(0 if ok else 1)
