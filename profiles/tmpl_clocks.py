import os, sys, ctypes
os.environ["SLPX_TAPE_JIT_CLOCKS"] = "1"
os.environ["SLPX_TAPE_JIT_VERBOSE"] = "1"
import numpy as np
import sleipnir_amd as sa
from tests.support import models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pp = models.cart_pole(N, 5.0 / N)
system = sa.System(pp)
L = sa.lib()
L.slpx_debug_tmpl_clocks.restype = ctypes.c_int
L.slpx_debug_tmpl_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
x0 = pp.get_x(); n, me, mi = pp.dims
system.set_state(x0[None], np.ones((1, mi)), np.zeros((1, me)), np.ones((1, mi)), np.array([0.1]))
for rep in range(3):
    system.sweep(True)
    system.sync()
NB = 2048
out = np.zeros(8 * NB, dtype=np.uint64)
nt = L.slpx_debug_tmpl_clocks(system._h, out.ctypes.data, NB)
c8 = out.reshape(NB, 8).astype(np.int64)
c = c8[:, :2]
used = c[:, 0] > 0
t0 = c[used, 0].min()
st = (c[:, 0] - t0) / 100.0; en = (c[:, 1] - t0) / 100.0   # us (100 MHz)
nb = used.sum()
print("template blocks", nt, "blocks recorded", nb, "kernel span us", en[used].max())
# per chunk of blocks: start range, end range, duration
def show(lo, hi, label):
    if hi <= lo:
        return
    s, e = st[lo:hi], en[lo:hi]
    print(f"{label:28s} [{lo:4d},{hi:4d}) start {s.min():6.2f}..{s.max():6.2f} end {e.min():6.2f}..{e.max():6.2f} dur {(e-s).min():6.2f}..{(e-s).max():6.2f}")
show(0, nt, "all template blocks")
# first family: 16 instance-chunks x 7 groups (block = chunk*ng + g)
show(nt, nb, "interpreted tasks")
d = en - st
order = np.argsort(-en[:nb])[:8]
print("last finishers:", [(int(b), round(float(st[b]), 2), round(float(en[b]), 2)) for b in order])

for b in sorted(set(list(range(min(nt, 32))) + [nt - 1, nt, nb - 1])):
    if b < 0 or b >= NB or c8[b, 0] <= 0:
        continue
    r = (c8[b] - t0) / 100.0
    print(f"block {b}: entry {r[0]:.2f} body {r[2]:.2f} leaves {r[3]:.2f} forward {r[4]:.2f} exit {r[1]:.2f}")
