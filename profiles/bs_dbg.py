import sys, numpy as np
sys.path.insert(0, "/root/repo")
import sleipnir_amd as sa
from tests.support import cases
from tests.support import models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
case = sys.argv[2] if len(sys.argv) > 2 else "step0"
sa.lib().slpx_graph_reset()
pp = models.cart_pole(N, 5.0 / N)
sy = sa.System(pp, batch=1, device=0)
n, me, mi = sy.info["n"], sy.info["m_e"], sy.info["m_i"]
x, s, y, z, mu = cases.newton_state(case, pp.get_x(), n, me, mi, 1.0)
sy.set_state(x, s, y, z, np.array([mu]))
for it in range(3):
    sy.reset_regularization()
    sy.newton_step(True)
    p = sy.get("p")[0].copy(); ps = sy.get("p_s")[0].copy(); pz = sy.get("p_z")[0].copy()
    sy.backsub()
    ps2 = sy.get("p_s")[0].copy(); pz2 = sy.get("p_z")[0].copy()
    bad = np.nonzero(~(np.abs(ps - ps2) <= 1e-12 * (1 + np.abs(ps2))))[0]
    print(it, "p finite", np.isfinite(p).all(), "ps mismatch rows", bad[:10], len(bad), "nan pz", int(np.isnan(pz).sum()))
    if len(bad):
        print("  fused", ps[bad[:5]], "standalone", ps2[bad[:5]])
