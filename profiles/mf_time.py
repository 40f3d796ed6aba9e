import sys, numpy as np
sys.path.insert(0, "/root/repo")
import sleipnir_amd as sa
from tests.support import cases
for N in (1000, 300, 500):
    sa.lib().slpx_graph_reset()
    pp = sa.Problem.cart_pole(N, 5.0 / N)
    sy = sa.System(pp, batch=1, device=0)
    n, me, mi = sy.info["n"], sy.info["m_e"], sy.info["m_i"]
    x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
    sy.set_state(x, s, y, z, np.array([mu]))
    for _ in range(3):
        t = sy.time_fused_step(200)
    print(N, t)
    sy.close(); pp.close()
