import sys, time
sys.path.insert(0, '.')
import sleipnir_amd as sa
from tests.support import models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for rep in range(3):
    sa.lib().slpx_graph_reset()
    pp = models.cart_pole(N, 5.0 / N)
    t = time.perf_counter()
    st, rep_ = pp.solve()
    print(N, st, rep_['iterations'], rep_['factorizations'], 't_total', rep_['t_total'], 'wall', time.perf_counter() - t,
          {k: round(rep_[k], 4) for k in ('t_setup', 't_kkt_build', 't_kkt_decomp', 't_kkt_solve', 't_line_search', 't_ad_refresh')})
    pp.close()
