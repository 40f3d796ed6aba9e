#!/usr/bin/env python3
"""The last of K chained steps as ONE timeline: the sweep's blocks (the generated kernel's own clocks,
SLPX_TAPE_JIT_CLOCKS) and the step kernel's tasks (a library built with -DSLPX_MF_CLOCKS) on the same 100 MHz
counter — when the sweep arrived, how long it waited for the step kernel before it, when the step kernel arrived and
when it saw the sweep.  profiles/ldlt_clocks.sh builds the library.

    SLPX_LIB=build/clocks_lib/libslpx.so LD_LIBRARY_PATH=build/clocks_lib PYTHONPATH=$PWD python profiles/chain_timeline.py [N] [steps]
"""
import ctypes
import os
import sys
import time

import numpy as np

os.environ["SLPX_TAPE_JIT_CLOCKS"] = "1"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300

import sleipnir_amd as sa  # noqa: E402
from tests.support import cases, models  # noqa: E402

L = sa.lib()
L.slpx_debug_ldlt_clocks.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
L.slpx_debug_tmpl_clocks.restype = ctypes.c_int
L.slpx_debug_tmpl_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
L.slpx_graph_reset()
pp = models.cart_pole(N, 5.0 / N)
sy = sa.System(pp, batch=1, device=0)
info = sy.info
n, me, mi = info["n"], info["m_e"], info["m_i"]
x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
sy.set_state(x, s, y, z, np.array([mu]))
sy.newton_steps(50)
sy.sync()
t_begin = time.perf_counter()
sy.newton_steps(STEPS)
sy.sync()
period = 1e6 * (time.perf_counter() - t_begin) / STEPS
T = info["ldlt_tasks"]
C = np.zeros((T, 24), dtype=np.uint64)
row = np.zeros(24, dtype=np.uint64)
for t in range(T):
    L.slpx_debug_ldlt_clocks(sy._h, 0xffff0000 | t, row.ctypes.data)
    C[t] = row
C = C.astype(np.int64)
NB = 2048
out = np.zeros(8 * NB, dtype=np.uint64)
nt = L.slpx_debug_tmpl_clocks(sy._h, out.ctypes.data, NB)
S = out.reshape(NB, 8).astype(np.int64)
S = S[S[:, 0] > 0]
body = S[S[:, 2] > 0][:, 2]  # (the generated bodies' blocks; an interpreted task has no such clock)
t0 = S[:, 0].min()
us = lambda v: (v - t0) / 100.0  # noqa: E731
print(f"cart-pole N={N}: {STEPS} chained steps, {period:.2f} us a step by the host's clock (both kernels instrumented); the last step, us after its sweep's first block came in")
print(f"  sweep ({len(S)} blocks): came in {us(S[:, 0].min()):6.2f} .. {us(S[:, 0].max()):6.2f}   wait over, plan read {us(body.min()):6.2f} .. {us(body.max()):6.2f}"
      f"   out {us(S[:, 1].min()):6.2f} .. {us(S[:, 1].max()):6.2f}")
print(f"  step kernel ({T} tasks): came in {us(C[:, 0].min()):6.2f} .. {us(C[:, 0].max()):6.2f}   image staged {us(C[:, 1].min()):6.2f} .. {us(C[:, 1].max()):6.2f}"
      f"   sweep seen {us(C[:, 14].min()):6.2f} .. {us(C[:, 14].max()):6.2f}   through {us(C[:, 16].min()):6.2f} .. {us(C[:, 16].max()):6.2f}")
print(f"  the sweep's first block waited {us(body.min()) - us(S[:, 0].min()):.2f} us; the step kernel saw it {us(C[:, 14].min()) - us(S[:, 1].max()):.2f} us after its last block;"
      f" from there to the last task {us(C[:, 16].max()) - us(C[:, 14].min()):.2f} us; in steady state the step kernel before this one was through at about"
      f" {us(C[:, 16].max()) - period:.2f}")
