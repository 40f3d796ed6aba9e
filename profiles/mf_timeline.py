#!/usr/bin/env python3
"""One launch of the step kernel of the fronts as a timeline of ALL its tasks (a library built with -DSLPX_MF_CLOCKS:
profiles/ldlt_clocks.sh builds it; every task keeps its phase clocks in LDS and writes them out when it is through).
Times in us after the first workgroup's entry; per round the earliest / median / latest task for every phase.

    SLPX_LIB=build/clocks_lib/libslpx.so LD_LIBRARY_PATH=build/clocks_lib PYTHONPATH=$PWD python profiles/mf_timeline.py [N] [chained 0|1]
"""
import ctypes
import os
import sys

import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
CHAINED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not CHAINED:
    os.environ["SLPX_CHAIN_TAPE"] = "0"

import sleipnir_amd as sa  # noqa: E402
from tests.support import cases, models  # noqa: E402

L = sa.lib()
L.slpx_debug_ldlt_clocks.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
L.slpx_graph_reset()
pp = models.cart_pole(N, 5.0 / N)
sy = sa.System(pp, batch=1, device=0)
info = sy.info
n, me, mi = info["n"], info["m_e"], info["m_i"]
x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
sy.set_state(x, s, y, z, np.array([mu]))
for _ in range(4):
    sy.reset_regularization()
    sy.newton_step(True)
T = info["ldlt_tasks"]
C = np.zeros((T, 24), dtype=np.uint64)
row = np.zeros(24, dtype=np.uint64)
for t in range(T):
    L.slpx_debug_ldlt_clocks(sy._h, 0xffff0000 | t, row.ctypes.data)
    C[t] = row
C = C.astype(np.int64)
t0 = C[:, 0].min()
us = (C - t0) / 100.0
# rounds: the leaf round is the big one; tasks are sorted by round
from tests.support import hostcheck  # noqa: E402
os.environ.setdefault("SLPX_LDLT_MF", "1")
F = hostcheck.HostCheck(pp).mf_fronts()
round_of = np.zeros(T, dtype=int)
for t in range(T):
    round_of[t] = F[F[:, 0] == t][0, 1]
names = [(0, "entry"), (1, "image staged"), (14, "sweep seen"), (15, "values + products in LDS"), (13, "sums done (all waves)"),
         (2, "values ready (update slots taken)"), (3, "levels done"), (5, "results out / counted"), (17, "row operands fetched"),
         (18, "ancestors' x in"), (19, "backward levels done"), (20, "x and directions out"), (16, "through")]
print(f"cart-pole N={N}, {'chained' if CHAINED else 'one step kernel after its sweep'}: {T} tasks; us after the first entry: earliest / median / latest (latest task)")
for r in sorted(set(round_of)):
    idx = np.where(round_of == r)[0]
    print(f"round {r}: {len(idx)} tasks")
    for k, name in names:
        v = us[idx, k]
        ok = C[idx, k] > 0
        if not ok.any():
            continue
        v = v[ok]
        print(f"   {name:36s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}   (task {int(idx[ok][np.argmax(v)])})")
