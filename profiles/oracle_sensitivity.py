#!/usr/bin/env python3
"""How much the REFERENCE ALGORITHM's own whole-solve outcome on the cart-pole swing-up depends on
the last bits of its input: the oracle (CPU restatement of interior_point.hpp) from the benchmark's
initial guess multiplied by (1 + 1e-13 u), u uniform in [-1, 1], six seeds per horizon.
Output of this script on the build container: profiles/r02_oracle_sensitivity.txt — at N=500 the
exit status itself changes (SUCCESS / LOCALLY_INFEASIBLE / FACTORIZATION_FAILED), which is why the
GPU tier pins whole-solve statuses only at the horizons where they are stable (tests/
test_restoration_gpu.py).  CPU only:  PYTHONPATH=$PWD python profiles/oracle_sensitivity.py"""
import sys, numpy as np
from tests.support import oracle
for N in (50, 500):
    for k in range(6):
        oracle.lib().orc_reset()
        op=oracle.OracleProblem.cart_pole(N,5.0/N)
        x=op.get_x()
        rng=np.random.default_rng(k)
        if k: x=x*(1+1e-13*rng.uniform(-1,1,len(x)))
        op.set_x(x)
        st,stats=op.solve()
        print(N,k,"status",st,"iterations",int(stats["iterations"]),flush=True)
