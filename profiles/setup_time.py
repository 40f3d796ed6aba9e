"""Time to the first System of a model in a fresh process (SLPX_SETUP_TIMING=1 prints the phases):
N = 1000 has a specialized tape kernel in sleipnir_amd/jit_cache; N = 700 has none and runs the
family's generic code object — neither waits for hipRTC.
    PYTHONPATH=$PWD python profiles/setup_time.py"""
import os
import sys
import time

sys.path.insert(0, '/root/repo')
os.environ["SLPX_SETUP_TIMING"] = "1"
os.environ["SLPX_TAPE_JIT_VERBOSE"] = "1"
import sleipnir_amd as sa  # noqa: E402
from tests.support import models

print('host cores', os.cpu_count())
for N in ([int(a) for a in sys.argv[1:]] or (1000, 700)):
    sa.lib().slpx_graph_reset()
    t = time.time(); pp = models.cart_pole(N, 5.0 / N); print(f"model N={N}", time.time() - t)
    t = time.time(); s = sa.System(pp, 1, 0); print(f"system N={N}", time.time() - t)
    s.close()
    t = time.time(); s = sa.System(pp, 1, 0); print(f"system again N={N}", time.time() - t)
    s.close()
