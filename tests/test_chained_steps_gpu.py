"""GPU tier: the chained Newton step (DeviceNlp::sweep_full_for_step; SLPX_CHAIN_TAPE=0 turns it
off): from the second of consecutive steps on, the AD sweep runs on a stream of its own beside the step
kernel, the two ordered through words in memory instead of the stream.  Whatever the order in which the
two kernels' workgroups reach the chip, a step must see exactly the V of ITS sweep and nothing of the
next one's: the same bits as the unchained step, for back-to-back steps (nothing but launches on the
host between them), for steps with uploads and downloads in between, and across changes of state."""
import numpy as np
import pytest

from tests.support import cases

pytestmark = pytest.mark.gpu


def _run(slpx, pp, states, scales, chained, monkeypatch):
    if chained:
        monkeypatch.delenv("SLPX_CHAIN_TAPE", raising=False)
    else:
        monkeypatch.setenv("SLPX_CHAIN_TAPE", "0")
    system = slpx.System(pp, batch=1, device=0)
    out = []
    try:
        system.set_scaling(scales)
        for k in range(3 * len(states)):
            x, s, y, z, mu = states[k % len(states)]
            system.set_state(x, s, y, z, np.array([mu]))
            system.reset_regularization()
            assert np.all(system.newton_step(True) == 0)  # the first chained step after uploads
            out.append([system.get(w)[0].copy() for w in ("p", "p_s", "p_z")])
            assert np.all(system.newton_steps(7) == 0)     # back to back: each sweep waits for the kernel before it
            assert np.all(system.newton_step(True) == 0)
            out.append([system.get(w)[0].copy() for w in ("p", "p_s", "p_z", "V")])
    finally:
        system.close()
    return out


@pytest.mark.parametrize("kind,N", [("cart_pole", 100), ("cart_pole", 1000)])
def test_chained_steps_give_the_bits_of_the_unchained_ones(fresh, slpx, orc, monkeypatch, kind, N):
    pp, op = cases.build_pair(kind, N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    states = [cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + k) for k in range(3)]
    plain = _run(slpx, pp, states, scales, False, monkeypatch)
    chained = _run(slpx, pp, states, scales, True, monkeypatch)
    for a, b in zip(plain, chained):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    # (and the states do differ: the comparison is not of one repeated step)
    assert not np.array_equal(plain[0][0], plain[2][0])
