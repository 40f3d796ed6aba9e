#!/usr/bin/env python3
"""Phase clocks of the batch-interleaved factorization (ldlt_il_kernels.h: SLPX_IL_CLOCK, slots
8..13 of slpx_debug_ldlt_clocks): first task of every round, first group of 16 problems —
{plan in LDS, values + update blocks in, levels done, update blocks out, L written} in us since
the workgroup's entry, beside the launch durations rocprofv3 shows.

    PYTHONPATH=$PWD python profiles/il_clocks.py [N] [batch]
"""
import ctypes
import sys

import numpy as np

import sleipnir_amd as sa
from tests.support import cases
from tests.support import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = sa.lib()
L.slpx_debug_ldlt_clocks.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
L.slpx_graph_reset()
pp = models.cart_pole(N, 5.0 / N)
sy = sa.System(pp, batch=B, device=0)
info = sy.info
n, me, mi = info["n"], info["m_e"], info["m_i"]
st = [cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0, seed=cases.SEED + b) for b in range(B)]
sy.set_state(*(np.stack([s[k] for s in st]) for k in range(4)), np.array([s[4] for s in st]))
print({k: info[k] for k in ("ldlt_rounds", "ldlt_tasks", "etree_height", "ldlt_levels", "nnz_L", "ldlt_pairs")})
out = np.zeros(24, dtype=np.uint64)
L.slpx_debug_ldlt_clocks(sy._h, 0, out.ctypes.data)
for r in range(info["ldlt_rounds"]):
    for _ in range(3):
        sy.reset_regularization()
        sy.newton_step(True)
    L.slpx_debug_ldlt_clocks(sy._h, (r + 1) % info["ldlt_rounds"], out.ctypes.data)
    f = out[8:14].astype(np.int64)
    print(f"round {r}: plan staged / values in / levels / updates out / L written, us:",
          [round(float(v - f[0]) / 100.0, 2) for v in f[1:]])
