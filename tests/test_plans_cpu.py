"""CPU tier: the compiled device plans (tape tasks, KKT gather maps, LDLᵀ task/round
schedule) interpreted sequentially on the host must reproduce the oracle's Newton
step (interior_point.hpp:426-482 after :245-251).  Exercises tape_compiler.cpp,
nlp.cpp, kkt_plan.cpp and ldlt_symbolic.cpp without a GPU."""
import pytest

from tests.support import cases, parity


@pytest.mark.parametrize("kind,N", [("cart_pole", 4), ("cart_pole", 37), ("flywheel", 50)])
@pytest.mark.parametrize("case", ["step0", "interior"])
def test_plan_interpreter_matches_oracle(fresh, slpx, orc, hostcheck, kind, N, case):
    pp, op = cases.build_pair(kind, N, slpx, orc)
    assert pp.dims == op.dims
    assert pp.types() == op.types()
    hc = hostcheck.HostCheck(pp)
    parity.check_newton_step(hc, op, case)


def test_small_tasks_force_many_rounds(fresh, slpx, orc, hostcheck):
    """Tiny LDS budgets: many tape tasks, many LDLᵀ rounds, lots of cross-task
    contribution slots — results must not change."""
    pp, op = cases.build_pair("cart_pole", 24, slpx, orc)
    hc = hostcheck.HostCheck(pp, task_entries=64, small_lds_bytes=24 * 1024)
    assert hc.info["ldlt_rounds"] >= 3
    parity.check_newton_step(hc, op, "interior")
