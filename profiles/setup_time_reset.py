"""Setup of the same model again and again in one process, the graph reset in between (what bench.py's
`compile_and_upload_warm` times): per-thread tables of the setup passes are warm, the graph does not grow.
    PYTHONPATH=$PWD python profiles/setup_time_reset.py [N ...]"""
import os
import sys
import time

os.environ["SLPX_SETUP_TIMING"] = "1"
import sleipnir_amd as sa  # noqa: E402
from tests.support import models  # noqa: E402

for N in ([int(a) for a in sys.argv[1:]] or (1000,)):
    for rep in range(4):
        sa.lib().slpx_graph_reset()
        t = time.time()
        pp = models.cart_pole(N, 5.0 / N)
        tm = time.time() - t
        print(f"== N={N} creation {rep}", file=sys.stderr, flush=True)
        t = time.time()
        s = sa.System(pp, 1, 0)
        print(f"model N={N} {tm:.4f} system {time.time() - t:.4f}", flush=True)
        print(f"model N={N} {tm:.4f} system {time.time() - t:.4f}", file=sys.stderr, flush=True)
        s.close()
        pp.close()
