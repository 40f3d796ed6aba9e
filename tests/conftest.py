import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def slpx():
    import sleipnir_amd

    sleipnir_amd.lib()
    return sleipnir_amd


@pytest.fixture(scope="session")
def orc():
    from tests.support import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def hostcheck():
    from tests.support import hostcheck as hc

    hc.lib()
    return hc


@pytest.fixture()
def fresh(slpx, orc):
    """Empty expression arenas on both sides."""
    orc.lib().orc_reset()
    slpx.lib().slpx_graph_reset()
    yield
