"""CPU tier: the tape compiler's family front end (tape_compiler.cpp: compile_tape_families — SURVEY.md §8f N4, "compile
one stage, replicate") against the flat compiler (SLPX_TAPE_TEMPLATES=0) it stands in for.

The front end finds the structurally identical components of a model on the raw graph, puts ONE member of every
family through the flat compiler and instantiates the others from it by position.  It must give the program the flat
compiler gives — the same tasks, nodes, slots and edges, the same structure shared — and a sweep of either program
the same V TO THE BIT (every output is written by one task, and a task's arithmetic is its structure)."""
import numpy as np
import pytest

from tests.support import cases, gfold, hostcheck, model, models


def _cart_pole(slpx, N):
    slpx.lib().slpx_graph_reset()
    return models.cart_pole(N, 5.0 / N)


def _flywheel(slpx, N):
    slpx.lib().slpx_graph_reset()
    return models.flywheel(N, 5.0 / N)


def _gfold(slpx, N):
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    return gfold.build(mp, N).p


STAT_KEYS = ("tape_tasks", "tape_nodes", "tape_slots", "tape_edges", "tape_levels", "tape_slot_levels", "tape_global_tasks",
             "tape_large_tasks", "tape_shared_tasks", "nV")


@pytest.mark.parametrize("name,make", [
    ("cart_pole_40", lambda s: _cart_pole(s, 40)),
    ("cart_pole_300", lambda s: _cart_pole(s, 300)),
    ("flywheel_50", lambda s: _flywheel(s, 50)),
    ("gfold_30", lambda s: _gfold(s, 30)),
])
def test_families_give_the_flat_compilers_program(fresh, slpx, monkeypatch, name, make):
    """flat (SLPX_TAPE_TEMPLATES=0) against the family front end on the SAME graph (SLPX_HESSIAN_FAMILIES=0: the gradient
    tree of the whole Lagrangian, as the flat compiler needs it): the same program, the same V to the bit.  And the
    family-first Hessian (r06, nlp.cpp: the symbolic reverse pass on one member of every family, the other members'
    gradient expressions never built): the same patterns and layout (nV), the same values up to the order in which a
    few adjoint terms are added — the gradient expressions of a stage no longer carry the multiplier-only terms its
    neighbours' linear rows contributed, and its adjoints are accumulated in the stage's own order."""
    got = {}
    for mode, hess in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("SLPX_TAPE_TEMPLATES", mode)
        monkeypatch.setenv("SLPX_HESSIAN_FAMILIES", hess)
        pp = make(slpx)
        h = hostcheck.HostCheck(pp)
        n, me, mi = h.n, h.m_e, h.m_i
        rng = np.random.default_rng(cases.SEED)
        x = np.asarray(pp.get_x()) + 1e-2 * rng.uniform(-1, 1, n)
        y = rng.uniform(-1, 1, me)
        z = np.exp(rng.uniform(-2, 2, mi))
        h.set_scaling(np.exp(rng.uniform(-1, 1, 1 + me + mi)))
        got[mode + hess] = ({k: h.info[k] for k in STAT_KEYS}, h.sweep(x, y, z, full=True), h.sweep(x, y, z, full=False),
                            h.info["graph_nodes"] if "graph_nodes" in h.info else None)
        h.close()
    assert got["00"][0] == got["10"][0], (got["00"][0], got["10"][0])
    assert np.array_equal(got["00"][1], got["10"][1])
    assert np.array_equal(got["00"][2], got["10"][2])
    assert np.any(got["10"][1] != 0.0)
    # the family-first Hessian: the same layout, the same values to rounding
    assert got["11"][0]["nV"] == got["10"][0]["nV"]
    a, b = got["11"][1], got["10"][1]
    assert a.shape == b.shape
    assert np.array_equal(a == 0.0, b == 0.0)
    err = np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))
    assert err <= 1e-13, err
    head = 1 + me + mi  # f, c_e, c_i: the same tape either way (the rest of that V is whatever the sweep before left)
    assert np.array_equal(got["11"][2][:head], got["10"][2][:head])
    print(name, got["11"][0], "family-first Hessian vs gradient tree of the Lagrangian:", err)


def test_family_compile_time_follows_the_horizon_gently(fresh, slpx, monkeypatch):
    """What the front end is for: the tape compile of N = 1000 stages costs the passes over the raw
    graph and ONE stage through the flat compiler.  A loose bound a return to per-instance compiling cannot meet: the
    whole setup with families well under the flat one at N = 400 (measured: 0.048 s against 0.132 s)."""
    import time

    t = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SLPX_TAPE_TEMPLATES", mode)
        pp = _cart_pole(slpx, 400)
        t0 = time.perf_counter()
        h = hostcheck.HostCheck(pp)
        t[mode] = time.perf_counter() - t0
        h.close()
    print(f"setup (structure + tape + KKT plan + symbolic LDLT, one thread) at N=400: flat {t['0']:.3f} s, families {t['1']:.3f} s")
    assert t["1"] < t["0"] / 1.8
