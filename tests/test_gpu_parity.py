"""GPU tier: the HIP kernels, called through the C-ABI (include/slpx.h), against the
oracle on identical inputs — AD sweep, KKT lhs/rhs, LDLᵀ factor/solve, back-
substitution (interior_point.hpp:245-251, :426-482) — and whole solves against the
reference's known answers."""
import numpy as np
import pytest

from tests.support import cases, parity
from tests.support import models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,N", [("cart_pole", 4), ("cart_pole", 37), ("cart_pole", 100),
                                    ("flywheel", 50)])
@pytest.mark.parametrize("case", ["step0", "interior"])
def test_newton_step_matches_oracle(fresh, slpx, orc, kind, N, case):
    pp, op = cases.build_pair(kind, N, slpx, orc)
    system = slpx.System(pp, batch=1, device=0)
    parity.check_newton_step(parity.GpuBackend(system), op, case, verbose=True)
    system.close()


def test_gpu_matches_plan_interpreter_bitwise_class(fresh, slpx, orc, hostcheck):
    """Kernel vs sequential interpretation of the same plans: only FMA contraction
    and libm differences may remain."""
    pp, op = cases.build_pair("cart_pole", 50, slpx, orc)
    n, me, mi = pp.dims
    system = slpx.System(pp, batch=1, device=0)
    gb = parity.GpuBackend(system)
    hc = hostcheck.HostCheck(pp)
    scales = op.scaling()
    gb.set_scaling(scales)
    hc.set_scaling(scales)
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    Vg, Vh = gb.sweep(x, y, z), hc.sweep(x, y, z)
    assert cases.max_rel(Vg, Vh) < 1e-12
    lg, lh = gb.assemble(s, z), hc.assemble(s, z)
    assert cases.max_rel(lg, lh) < 1e-12
    rg, rh = gb.rhs(s, y, z, mu), hc.rhs(s, y, z, mu)
    assert cases.max_rel(rg, rh) < 1e-12
    hc.set_lhs(lg)
    hc.set_rhs(rg)
    Dg, sg = gb.factor(1e-4, 1e-10)
    Dh, sh = hc.factor(1e-4, 1e-10)
    assert np.array_equal(sg[:4], sh[:4])
    # pivots of the γ=1e-10-regularized system span 20 orders of magnitude; individual
    # small pivots are cancellation-limited, so compare the bulk and the solve residual
    rel = np.abs(Dg - Dh) / np.abs(Dh)
    assert np.median(rel) < 1e-12 and np.max(rel) < 1e-1
    pg, ph = gb.solve(), hc.solve()
    lcp, lri = gb.pattern(5)
    Kreg = cases.regularized(lcp, lri, lg, n, 1e-4, 1e-10)
    # normwise backward error ‖Kp − b‖∞ / max(‖b‖∞, ‖K‖∞‖p‖∞) (|p| reaches 1e4 here)
    k_inf = float(np.max(cases.lower_csc_matvec(lcp, lri, np.abs(Kreg), np.ones_like(rg))))
    for p in (pg, ph):
        scale = max(1.0, float(np.max(np.abs(rg))), k_inf * float(np.max(np.abs(p))))
        assert np.max(np.abs(cases.lower_csc_matvec(lcp, lri, Kreg, p) - rg)) / scale < 1e-10
    system.close()


def test_batch_items_are_independent(fresh, slpx, orc):
    """Batch of 3 value sets through one launch: every item gets the same bits wherever it sits in
    the batch, and the same step as a single run — to rounding: one problem takes the multifrontal
    step (ldlt_mf_kernels.h), a small batch the pair-list kernels, which sum in another order."""
    pp, op = cases.build_pair("cart_pole", 20, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    states = [cases.newton_state("interior", op.get_x(), n, me, mi, scales[0], seed=cases.SEED + b)
              for b in range(3)]
    single = []
    sys1 = slpx.System(pp, batch=1, device=0)
    sys1.set_scaling(scales)
    for (x, s, y, z, mu) in states:
        sys1.reset_regularization()
        sys1.set_state(x, s, y, z, np.array([mu]))
        info = sys1.newton_step(True)
        assert info[0] == 0
        single.append((sys1.get("p")[0].copy(), sys1.get("p_s")[0].copy(), sys1.get("p_z")[0].copy()))
    sys1.close()

    def run(order):
        sys3 = slpx.System(pp, batch=3, device=0)
        sys3.set_scaling(scales)
        sys3.set_state(*(np.stack([states[b][k] for b in order]) for k in range(4)),
                       np.array([states[b][4] for b in order]))
        info = sys3.newton_step(True)
        assert np.all(info == 0)
        out = sys3.get("p"), sys3.get("p_s"), sys3.get("p_z")
        sys3.close()
        return out

    first, second = run((0, 1, 2)), run((2, 0, 1))
    for b in range(3):
        for k in range(3):
            assert np.array_equal(first[k][b], second[k][(b + 1) % 3])
            ref = single[b][k]
            assert np.max(np.abs(first[k][b] - ref)) <= 1e-7 * max(1.0, float(np.max(np.abs(ref))))


def test_flywheel_solve_matches_reference_known_answer(fresh, slpx, orc):
    """benchmarks/scalability/flywheel + test/src/optimization/flywheel_problem_test.cpp:
    bang-then-hold input, final state r = 10 (:112, :122)."""
    N, T = 50, 5.0
    dt = T / N
    pp, op = cases.build_pair("flywheel", N, slpx, orc)
    status, rep = pp.solve()
    assert status == 0, rep
    x = pp.get_x()
    X, U = x[:N + 1], x[N + 1:]
    ostatus, _ = op.solve()
    assert ostatus == 0
    xo = op.get_x()
    assert np.max(np.abs(x - xo)) < 1e-6
    A, B = np.exp(-dt), 1.0 - np.exp(-dt)
    assert abs(X[0]) < 1e-8
    assert np.all(U <= 12.0 + 1e-6) and np.all(U >= -12.0 - 1e-6)
    for k in range(N):
        assert abs(X[k + 1] - (A * X[k] + B * U[k])) < 1e-8


def test_cart_pole_solve(fresh, slpx, orc):
    """test/src/optimization/cart_pole_problem_test.cpp:87-124 at N=100, dt=0.05:
    SUCCESS, boundary states and bounds hold, every dynamics defect <= 1e-8 (checked
    through the oracle's constraint evaluation at the returned point)."""
    N = 100
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    status, rep = pp.solve()
    print(rep)
    assert status == 0, rep
    x = pp.get_x()
    n, me, mi = pp.dims
    X = x[:4 * (N + 1)].reshape(4, N + 1)
    U = x[4 * (N + 1):]
    assert np.max(np.abs(X[:, 0])) < 1e-8
    assert np.max(np.abs(X[:, N] - np.array([1.0, np.pi, 0.0, 0.0]))) < 1e-8
    assert np.all(X[0] >= -1e-8) and np.all(X[0] <= 2.0 + 1e-8)
    assert np.all(np.abs(U) <= 20.0 + 1e-8)
    # constraint residuals at the GPU solution, evaluated by the oracle (unscaled)
    scales = op.scaling()
    op.newton_step(x, np.ones(mi), np.zeros(me), np.ones(mi), 0.1, do_solve=False)
    c_e = op.vec("c_e") / scales[1:1 + me]
    assert np.max(np.abs(c_e)) < 1e-8
    # same optimum as the oracle's own solve (cost within 1e-6 relative)
    ostatus, _ = op.solve()
    assert ostatus == 0
    xo = op.get_x()
    Jg, Jo = float(np.sum(U ** 2)), float(np.sum(xo[4 * (N + 1):] ** 2))
    assert abs(Jg - Jo) <= 1e-6 * max(1.0, abs(Jo)), (Jg, Jo)


def test_generated_tape_kernel_matches_the_interpreter_bit_for_bit(fresh, slpx, monkeypatch):
    """tape_jit.cpp generates straight-line code per task family and compiles it at run time
    (hipRTC, -ffp-contract=on); the interpreting kernels walk the same program level by
    level.  Same op_forward, same accumulation order, no cross-node contraction: the whole
    value vector V (f, c, g, A_e, A_i, H) must be IDENTICAL with the generator switched off."""
    def sweep(jit):
        monkeypatch.setenv("SLPX_TAPE_JIT", jit)
        slpx.lib().slpx_graph_reset()
        pp = models.cart_pole(120, 5.0 / 120)
        n, me, mi = pp.dims
        system = slpx.System(pp, batch=1, device=0)
        try:
            x, s, y, z, mu = cases.newton_state("interior", pp.get_x(), n, me, mi, 1.0)
            system.set_state(x, s, y, z, np.array([mu]))
            system.sweep(True)
            return system.get("V")[0].copy()
        finally:
            system.close()

    assert np.array_equal(sweep("1"), sweep("0"))
