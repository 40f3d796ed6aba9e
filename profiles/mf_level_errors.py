#!/usr/bin/env python3
"""The multifrontal step level by level, double against long double (tests/support/hostcheck.cpp: hc_mf_level_errors):
where along the elimination the rounding error of a Newton step's factorization and backward solve appears.
    SLPX_LDLT_MF=1 python profiles/mf_level_errors.py [N] [seed offset]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

import sleipnir_amd as slpx  # noqa: E402
from tests.support import cases, hostcheck, oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
orc.lib().orc_reset()
slpx.lib().slpx_graph_reset()
pp, op = cases.build_pair("cart_pole", N, slpx, orc)
n, me, mi = pp.dims
hc = hostcheck.HostCheck(pp)
scales = op.scaling()
hc.set_scaling(scales)
x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + seed)
info, _ = op.newton_step(x, s, y, z, mu, True, hc.perm())
delta, gamma, _, _ = op.reg()
hc.sweep(x, y, z, True)
hc.assemble(s, z)
hc.rhs(s, y, z, mu)
rows = hc.mf_level_errors(delta, gamma)
print(f"# cart-pole N={N}, interior state seed {cases.SEED + seed}, delta {delta:g} gamma {gamma:g}; the plan's fronts in double against")
print("# long double (eps 5.4e-20), same order of operations; a row = one level of one round, all its tasks together")
print("# phase        round level  values  worst front (max|d-q| / max|q|)  componentwise median   componentwise max")
for ph, r, l, cnt, wn, med, mx in rows:
    print(f"  {'factorization' if ph == 0 else 'backward solve':<14s} {int(r):3d} {int(l):5d} {int(cnt):7d}  {wn:28.2e}  {med:20.2e}  {mx:18.2e}")
