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


@pytest.mark.parametrize("kind,N", [("cart_pole", 37), ("flywheel", 50)])
def test_rhs_row_of_the_factorization_is_the_forward_solve(fresh, slpx, orc, hostcheck, kind, N):
    """The factorization carries the rhs as an extra row (ldlt_symbolic.cpp); what it leaves
    behind must be z = D^-1 L^-1 P b, i.e. backward substitution alone reproduces the full
    solve — the path the device takes inside a Newton step."""
    pp, op = cases.build_pair(kind, N, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    n, me, mi = hc.n, hc.m_e, hc.m_i
    scales = op.scaling()
    hc.set_scaling(scales)
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    hc.sweep(x, y, z, True)
    hc.assemble(s, z)
    hc.rhs(s, y, z, mu)
    hc.factor(1e-4, 1e-10)
    p_full = hc.solve()
    p_fused = hc.solve_after_factor()
    assert cases.max_rel(p_fused, p_full) <= 1e-9
    hc.close()


def test_generated_tape_kernel_compiles_for_gfx950(fresh, slpx, orc, hostcheck):
    """tape_jit.cpp: the run-time generated tape kernel (straight-line bodies for the task
    families + the interpreter for the rest, prelude = tape_ops.h / tape_device.h /
    tape_interp.h as text) must compile with hipRTC for gfx950 — which needs no GPU, so a
    broken generator or prelude is caught on the CPU tier."""
    pp, _ = cases.build_pair("cart_pole", 64, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    # the stage family (64 members) and the cost groups qualify for a body
    assert hc.tape_jit_compiles() >= 1


def test_no_supernode_reaches_an_mfma_tile(fresh, slpx, orc, hostcheck):
    """north_star: "MFMA only on dense supernode panels".  v_mfma_f64_16x16x4_f64 wants panels
    at least 16 columns wide; with the depth-first ordering used here (nested dissection,
    8-node leaves) the factors of the transcription problems have no such supernode at all —
    the measured reason the panel path is not built.  Recorded for cart-pole
    N=1000 (the BASELINE horizon) and for g-fold, whose inequality rows give dense blocks."""
    from tests.support import gfold, model

    pp, _ = cases.build_pair("cart_pole", 1000, slpx, orc)
    sn = hostcheck.HostCheck(pp).supernodes()
    assert sn["widest"] < 16 and sn["cols_in_ge16"] == 0
    assert sn["longest_column"] < 16
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    g = gfold.build(mp, 100)
    sg = hostcheck.HostCheck(g.p).supernodes()
    assert sg["widest"] < 16 and sg["cols_in_ge16"] == 0
    print("supernodes cart-pole N=1000:", sn, " g-fold N=100:", sg)
