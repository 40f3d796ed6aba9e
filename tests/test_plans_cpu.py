"""CPU tier: the compiled device plans (tape tasks, KKT gather maps, LDLᵀ task/round
schedule) interpreted sequentially on the host must reproduce the oracle's Newton
step (interior_point.hpp:426-482 after :245-251).  Exercises tape_compiler.cpp,
nlp.cpp, kkt_plan.cpp and ldlt_symbolic.cpp without a GPU."""
import pytest

from tests.support import cases, parity
from tests.support import models


@pytest.mark.parametrize("kind,N", [("cart_pole", 4), ("cart_pole", 37), ("flywheel", 50)])
@pytest.mark.parametrize("case", ["step0", "interior"])
def test_plan_interpreter_matches_oracle(fresh, slpx, orc, hostcheck, kind, N, case):
    pp, op = cases.build_pair(kind, N, slpx, orc)
    assert pp.dims == op.dims
    assert pp.types() == op.types()
    hc = hostcheck.HostCheck(pp)
    parity.check_newton_step(hc, op, case)


@pytest.mark.parametrize("env", [{"SLPX_SUPERNODAL": "0"}, {"SLPX_SN_MIN_WIDTH": "2"}, {"SLPX_SN_MIN_WIDTH": "3"}])
@pytest.mark.parametrize("kind,N", [("cart_pole", 37), ("flywheel", 50)])
def test_plan_interpreter_with_other_supernode_settings(fresh, slpx, orc, hostcheck, monkeypatch, env, kind, N):
    """Column levels (no supernodes) and chains of every width from 2 / 3 up: the same step."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pp, op = cases.build_pair(kind, N, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    plan = hc.supernode_plan()
    if env.get("SLPX_SUPERNODAL") == "0":
        assert plan["widest"] == 1 and plan["count"] == hc.n + hc.m_e
    else:
        assert plan["widest"] >= 2 and min(w for w in plan["width_hist"] if w > 1) >= int(env["SLPX_SN_MIN_WIDTH"])
    parity.check_newton_step(hc, op, "interior")


def relaxed_chains(parent, cc):
    """Maximal chains j -> parent(j) with |struct(L_j)| = |struct(L_parent)| + 1 (equal structure up
    to the parent itself), any number of other children: what ldlt_symbolic.cpp amalgamates before
    it cuts chains at kSnWidthMax columns.  Returns the list of (width, rows below the chain)."""
    n = len(parent)
    taken = [False] * n
    out = []
    for j in range(n):
        if taken[j]:
            continue
        k, w = j, 1
        taken[j] = True
        while parent[k] >= 0 and cc[k] == cc[parent[k]] + 1 and not taken[parent[k]]:
            k = parent[k]
            taken[k] = True
            w += 1
        out.append((w, int(cc[k])))
    return out


def test_supernodal_levels(fresh, slpx, orc, hostcheck):
    """relaxed supernodes (separator cliques and chains, whatever the number of
    children) cut the levels on the critical path of the factorization from the etree height to
    about a third; the width histogram is the measured basis of the MFMA decision:
    cart-pole chains are at most 13 columns wide (one root chain), g-fold has panels of 16-26
    columns over 8-20 rows."""
    from tests.support import gfold, model

    pp, _ = cases.build_pair("cart_pole", 1000, slpx, orc)
    hc = hostcheck.HostCheck(pp)
    plan = hc.supernode_plan()
    assert hc.info["etree_height"] == 50
    assert plan["critical_levels"] <= 22
    assert 4 <= plan["widest"] <= 8          # chains are cut at kSnWidthMax = 8 columns, shorter than 4 stay columns
    chains = relaxed_chains(*hc.ldlt_tree())
    widths = sorted(w for w, _ in chains)
    assert widths[-1] == 13 and sum(1 for w in widths if w >= 12) == 1
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    g = gfold.build(mp, 100)
    hg = hostcheck.HostCheck(g.p)
    pg = hg.supernode_plan()
    assert hg.info["etree_height"] >= 60 and pg["critical_levels"] <= 36
    gch = relaxed_chains(*hg.ldlt_tree())
    wide = sorted((w, r) for w, r in gch if w >= 12)
    assert wide and wide[-1][0] >= 16        # dense panels an MFMA tile could bite on exist here
    print("cart-pole N=1000 plan:", plan, " g-fold N=100 plan:", pg, " g-fold chains >= 12 columns (w, rows below):", wide)


@pytest.mark.parametrize("width,levels_cp,levels_gf", [(16, 18, 29), (32, 18, 24)])
def test_wider_supernodes_what_if(fresh, slpx, orc, hostcheck, monkeypatch, width, levels_cp, levels_gf):
    """DESIGN.md §4 (MFMA update blocks: built in r03 — ldlt_mf_kernels.h: mf_update_mfma — and tested in the GPU tier): what cutting chains at 16 / 32 columns
    instead of the kernels' 8 would do to the critical path — the plans are valid (the host
    interpreter reproduces the oracle's step with them), the device kernels do not take them."""
    from tests.support import gfold, model

    monkeypatch.setenv("SLPX_HOSTCHECK_SN_MAX_WIDTH", str(width))
    pp, op = cases.build_pair("cart_pole", 1000, slpx, orc)
    plan = hostcheck.HostCheck(pp).supernode_plan()
    assert plan["critical_levels"] == levels_cp and plan["widest"] == 13
    mp = model.Model(model.ProductBackend("hostcheck"))
    mp.be.reset()
    pg = hostcheck.HostCheck(gfold.build(mp, 100).p).supernode_plan()
    assert pg["critical_levels"] == levels_gf and pg["widest"] == min(width, 20)
    pq, oq = cases.build_pair("cart_pole", 37, slpx, orc)
    parity.check_newton_step(hostcheck.HostCheck(pq), oq, "interior")


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
    """north_star: "MFMA only on dense supernode panels".  A 16-column panel for
    v_mfma_f64_16x16x4_f64 does not occur: with the depth-first ordering used here (nested
    dissection, 8-node leaves) the factors of the transcription problems have no supernode that
    wide.  Recorded for cart-pole N=1000 (the BASELINE horizon) and for g-fold, whose inequality
    rows give dense blocks.  (r03: where the matrix cores do get work is the UPDATE BLOCK of a
    front — rows x rows below the pivots, four pivot columns per instruction — see
    tests/test_multifrontal_cpu.py::test_gfold_fronts_reach_the_matrix_cores and DESIGN.md §4.)"""
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


def test_generated_kernel_of_a_family_serves_every_horizon(fresh, slpx, tmp_path):
    """tape_jit.cpp (ParamSink): in the GENERIC source the numbers that follow the horizon — base
    indices, constants derived from the timestep — are members of a table in device memory, so
    two horizons of one model generate the same text (one code object, no hipRTC wait for a new
    N); the SPECIALIZED source has them as literals and differs.  hipRTC cross-compiles both."""
    names = {}
    for N in (16, 24):
        slpx.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        assert pp.prebuild_kernels(tmp_path / f"N{N}") >= 1
        pp.close()
        names[N] = {f.name for f in (tmp_path / f"N{N}").iterdir()}
    shared = names[16] & names[24]
    assert shared, "no code object in common: the generic source depends on the horizon"
    assert names[16] - shared and names[24] - shared  # the specialized ones
    # one generic beside every specialized — but for a tape whose 256-thread interpreted tasks exist at one
    # horizon and not at the other: its generic kernel has workgroups of 256 there (those tasks ride in
    # its launch, tape_jit.cpp: block_threads) and of 64 here, two code objects for the family
    n_programs = len(names[16]) // 2
    assert len(names[16]) == 2 * n_programs and n_programs - 1 <= len(shared) <= n_programs


@pytest.mark.parametrize("kind,N", [("cart_pole", 60), ("cart_pole", 700), ("flywheel", 50)])
def test_plans_do_not_depend_on_the_thread_count(kind, N):
    """The setup passes run in chunks on a pool of threads (csrc/setup_threads.hpp), the two sides of every
    nested-dissection separator on two: structure, both tapes, KKT plan, LDLT plan and the fronts come out the
    same to the byte with one thread, three and eight (hashes of every plan array, hostcheck.cpp: hc_plan_hash)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    seen = {}
    for threads in ("1", "3", "8"):
        env = dict(os.environ, SLPX_SETUP_THREADS=threads, SLPX_LDLT_MF="1", PYTHONPATH=str(root))
        out = subprocess.run([sys.executable, "-m", "tests.support.plan_hash_cli", kind, str(N)], cwd=root, env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        seen[threads] = json.loads(out.stdout.strip().splitlines()[-1])
    assert seen["1"] == seen["3"] == seen["8"], seen
