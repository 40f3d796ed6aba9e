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
WITHIN = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # which task of each round records (0: its first)
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
L.slpx_debug_ldlt_clocks(sy._h, 0 | (WITHIN << 8), out.ctypes.data)
last_entry = -1
for r in range(info["ldlt_rounds"]):
    for _ in range(3):
        sy.reset_regularization()
        sy.newton_step(True)
    L.slpx_debug_ldlt_clocks(sy._h, ((r + 1) % info["ldlt_rounds"]) | (WITHIN << 8), out.ctypes.data)
    f = out[0:6].astype(np.int64)
    if info.get("ldlt_multifrontal") and int(f[0]) == 0:
        sys.exit("this library carries no clocks in the step kernel of the fronts: bash profiles/ldlt_clocks.sh builds one that does")
    if r > 0 and int(f[0]) == last_entry:
        continue  # (the round has no task number WITHIN: nothing was recorded)
    last_entry = int(f[0])
    b = out[16:21].astype(np.int64)
    if info.get("ldlt_multifrontal"):
        # (the step kernel of the fronts: the end of every level, us after the values were in / after the ancestors' x)
        fl = [int(v) for v in out[8:13] if f[2] < int(v) <= f[3]]
        print(f"   round {r} task +{WITHIN}: wave 0 through its sums {(int(out[4]) - int(f[0])) / 100.0:.2f}, every wave {(int(out[13]) - int(f[0])) / 100.0:.2f} us after entry")
        print(f"   round {r} task +{WITHIN}: sweep seen {(int(out[14]) - int(f[0])) / 100.0:.2f}, values and products in LDS {(int(out[15]) - int(f[0])) / 100.0:.2f} us after entry")
        bl = [int(out[k]) for k in (6, 7, 21, 22, 23) if b[2] < int(out[k]) <= b[3]]
        print(f"   round {r} task +{WITHIN}: factorization levels end at", [round((v - int(f[2])) / 100.0, 2) for v in fl],
              "us after the values; backward levels at", [round((v - int(b[2])) / 100.0, 2) for v in bl], "us after the ancestors' x")
    merged = bool(info.get("ldlt_multifrontal")) or b[0] < f[0]  # one launch for both (ldlt_factor_solve_kernel): slot 16 is not written, the solve's clocks are on the factorization's axis
    base = f[0] if merged else b[0]
    f[4] = f[3]  # (slot 4 is a finer clock now)
    print(f"round {r}: factor staged/gathered/levels/updates/exit us:", [round(float(v - f[0]) / 100.0, 2) for v in f[1:]],
          (" solve (same launch, same axis) values-in-LDS/ancestors-folded/levels/exit us:" if merged else
           " bwd staged/gathered/levels/exit us:"), [round(float(v - base) / 100.0, 2) for v in b[1:]])
