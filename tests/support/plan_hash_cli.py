"""Prints the hashes of the compiled plans (hostcheck.HostCheck.plan_hash) of one model as JSON: run in a process
of its own so that SLPX_SETUP_THREADS / SLPX_LDLT_MF take effect (the setup pool is made once per process).
    python -m tests.support.plan_hash_cli cart_pole 300"""
import json
import sys

import sleipnir_amd as sa
from tests.support import hostcheck, models


def main(kind: str, N: int) -> None:
    sa.lib().slpx_graph_reset()
    pp = models.cart_pole(N, 5.0 / N) if kind == "cart_pole" else models.flywheel(N, 0.005)
    print(json.dumps(hostcheck.HostCheck(pp).plan_hash()))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
