#!/usr/bin/env python3
"""Phase clocks (wall_clock64, 100 MHz) of the first task of every LDLT round of a cart-pole
Newton step: factorization {staged, values gathered, levels done, update blocks done, exit} and
backward solve, in microseconds since the workgroup's entry (slpx_debug_ldlt_clocks).

    PYTHONPATH=$PWD python profiles/ldlt_clocks.py [N]
"""
import ctypes
import sys

import numpy as np

import sleipnir_amd as sa
from tests.support import cases
from tests.support import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = sa.lib()
L.slpx_debug_ldlt_clocks.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
L.slpx_graph_reset()
pp = models.cart_pole(N, 5.0 / N)
sy = sa.System(pp, batch=1, device=0)
info = sy.info
n, me, mi = info["n"], info["m_e"], info["m_i"]
x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
sy.set_state(x, s, y, z, np.array([mu]))
print({k: info[k] for k in ("ldlt_rounds", "ldlt_tasks", "etree_height", "ldlt_levels", "ldlt_supernodes", "nnz_L")})
out = np.zeros(24, dtype=np.uint64)
L.slpx_debug_ldlt_clocks(sy._h, 0, out.ctypes.data)
for r in range(info["ldlt_rounds"]):
    for _ in range(3):
        sy.reset_regularization()
        sy.newton_step(True)
    L.slpx_debug_ldlt_clocks(sy._h, (r + 1) % info["ldlt_rounds"], out.ctypes.data)
    f = out[0:6].astype(np.int64)
    b = out[16:21].astype(np.int64)
    print(f"   level loop cycles: pass A {int(out[6])}  pass B {int(out[7])}")
    merged = b[0] < f[0]  # one launch for both (ldlt_factor_solve_kernel): slot 16 is not written, the solve's clocks are on the factorization's axis
    base = f[0] if merged else b[0]
    print(f"round {r}: factor staged/gathered/levels/updates/exit us:", [round(float(v - f[0]) / 100.0, 2) for v in f[1:]],
          (" solve (same launch, same axis) values-in-LDS/ancestors-folded/levels/exit us:" if merged else
           " bwd staged/gathered/levels/exit us:"), [round(float(v - base) / 100.0, 2) for v in b[1:]])
