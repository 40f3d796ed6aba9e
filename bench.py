#!/usr/bin/env python3
"""Newton steps/s of the interior-point Newton step on the cart-pole direct-
transcription problem (BASELINE.json: N=1000, single problem per GPU, fp64).

One "step" = AD refresh {g, A_e, H} + KKT lhs/rhs assembly + regularized LDLᵀ
(inertia-correcting loop, every attempt a device factorization) + solve + (p_s, p_z)
back-substitution, i.e. interior_point.hpp:809-812 + :426-482, on inputs already
resident in HBM (the seeded interior state of SURVEY.md §8d).

  python bench.py --gpus N --steps K --warmup W [--workload single|batch512]

--gpus N > 1 without a torchrun environment re-executes this script under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); under a
launcher the flag must equal WORLD_SIZE.  Workloads:
  single    (default, BASELINE config 2) every rank steps its own replica of ONE N=1000
            problem — the path has no cross-problem exchange ("replicas only")
  gfold     (BASELINE config 5) the g-fold powered-descent OCP at N=100 (quadratic cone-type
            inequality rows: the general A_i^T Sigma A_i product, chains of 16-26 columns in L),
            built through the expression C-ABI by tests/support/gfold.py; replicas like `single`
  batch512  (BASELINE config 4) 512 independent cart-pole N=500 problems, problem b seeded
            with SEED + b, sharded contiguously over the ranks (sleipnir_amd.dist.shard_range:
            64 per GPU at 8 GPUs; multistart.hpp:45-74 hands whole solves to threads the same
            way); the per-problem {status, delta, gamma} table is all-gathered at the end.
The only collectives are the barrier, the MAX of the elapsed time and that gather.

The timed region is `--repeats` (default 5) blocks of EXACTLY K steps, each bracketed by
barrier + synchronize on both sides and MAX-reduced over the ranks; the line reports the
MEDIAN block (all block times are in `block_ms`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from tests.support import models

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_TAG = "r06"    # profiles/<tag>_traffic.json feeds roofline.traffic


def cpu_baseline(N: int, dt: float, budget_s: float = 15.0):
    """The oracle (CPU restatement of the reference path) timed on one host core."""
    from tests.support import cases, oracle

    oracle.lib().orc_reset()
    op = oracle.OracleProblem.cart_pole(N, dt)
    n, me, mi = op.dims
    scales = op.scaling()
    x, s, y, z, mu = cases.newton_state("interior", op.get_x(), n, me, mi, scales[0])
    op.newton_step(x, s, y, z, mu, True, None, False)  # includes analyzePattern
    t0 = time.perf_counter()
    steps = 0
    phases = np.zeros(4)
    while True:
        # 2 = keep the symbolic analysis, forget delta/gamma: every step starts like the first
        # iteration of a solve, as the GPU leg's reset_regularization() makes its steps do
        _, t = op.newton_step(x, s, y, z, mu, True, None, 2)
        phases += np.array([t["t_ad"], t["t_build"], t["t_decomp"], t["t_solve"]])
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 200:
            break
    el = time.perf_counter() - t0
    _, _, nfact, nnzL = op.reg()
    return {
        "value": steps / el, "unit": "Newton steps/s", "cores": 1, "kind": "port",
        "sample": f"{steps} steps of cart-pole N={N} (interior state), oracle/ single thread, "
                  f"regularization memory cleared before every step, {nfact} factorizations/step "
                  f"(the unregularized attempt, which ends on a zero pivot, and delta=1e-4), nnz(L)={nnzL}",
        "ms_per_step": 1e3 * el / steps,
        "phase_ms": {k: 1e3 * v / steps for k, v in
                     zip(["ad_refresh", "kkt_build", "kkt_decomp", "kkt_solve"], phases)},
    }


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_launcher(args) -> int:
    """`python bench.py --gpus N` outside a launcher: become N ranks (one per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())]
    cmd += sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def scaling_at_x0(system, info, x0, me, mi, B):
    """problem_scaling.hpp:100-107 from an unscaled device sweep at the initial guess."""
    system.set_state(np.tile(x0, (B, 1)), np.ones((B, mi)), np.zeros((B, me)), np.ones((B, mi)),
                     np.full(B, 0.1))
    system.sweep(True)
    V = system.get("V")[0]
    gmax = np.max(np.abs(V[info["off_g"]:info["off_Ae"]])) if info["nnz_g"] else 0.0
    scales = np.ones(1 + me + mi)
    with np.errstate(divide="ignore"):
        scales[0] = min(1.0, 100.0 / gmax) if gmax > 0 else 1.0
        for which, off, cnt, base in ((1, "off_Ae", "nnz_Ae", 1), (2, "off_Ai", "nnz_Ai", 1 + me)):
            cp, ri = system.pattern(which)
            rn = np.zeros(me if which == 1 else mi)
            np.maximum.at(rn, ri, np.abs(V[info[off]:info[off] + info[cnt]]))
            scales[base:base + len(rn)] = np.minimum(100.0 / rn, 1.0)
    return scales


def make_system(sa, cases, N, problem_ids, device, kind="cart_pole"):
    """Compiles cart-pole N (or g-fold N) for `device` with one value set per problem id (seeded
    interior states) resident in HBM."""
    dt = 5.0 / N
    sa.lib().slpx_graph_reset()
    t0 = time.perf_counter()
    if kind == "gfold":
        from tests.support import gfold, model

        pp = gfold.build(model.Model(model.ProductBackend("gpu")), N).p
    else:
        pp = models.cart_pole(N, dt)
    t_model = time.perf_counter() - t0
    B = len(problem_ids)
    t0 = time.perf_counter()
    system = sa.System(pp, batch=B, device=device)
    t_compile = time.perf_counter() - t0
    info = system.info
    n, me, mi = info["n"], info["m_e"], info["m_i"]
    x0 = pp.get_x()
    scales = scaling_at_x0(system, info, x0, me, mi, B)
    system.set_scaling(scales)
    st = [cases.newton_state("interior", x0, n, me, mi, scales[0], seed=cases.SEED + b) for b in problem_ids]
    system.set_state(np.stack([s_[0] for s_ in st]), np.stack([s_[1] for s_ in st]),
                     np.stack([s_[2] for s_ in st]), np.stack([s_[3] for s_ in st]),
                     np.array([s_[4] for s_ in st]))
    return pp, system, {"model": t_model, "compile_and_upload": t_compile}


def kernel_groups(system, iters):
    """Per-kernel-group launch durations (HIP events on the library's stream) and the
    algorithmic bytes of SURVEY.md §8d evaluated on the actual patterns."""
    info = system.info
    system.reset_regularization()
    kt = system.time_step(iters=iters, refresh_ad=True)
    nf = max(1.0, kt["factorizations"])
    return kt, nf, {
        # SURVEY.md §8d's AD-refresh figure: read 8 (n + m_e + m_i), write 8 x (nnz of the outputs),
        # per problem.  (r02 added the interpreted tape program's bytes here and multiplied them by
        # the batch: the generated kernel has the program compiled in, and a batch shares whatever
        # is left of it — VERDICT r02 item 9.)
        "tape_sweep": (kt["sweep"], info["sweep_bytes"]),
        "kkt_assemble": (kt["assemble"], info["assemble_bytes"]),
        "kkt_rhs": (kt["rhs"], info["rhs_bytes"]),
        "ldlt_factor": (kt["factor"] / nf, info["factor_bytes"]),
        # the timed launch is the BACKWARD substitution only (the forward one rides in the
        # factorization): half of the 32 l + 16(n + m_e) of SURVEY.md §8d
        "ldlt_solve": (kt["solve"], info["solve_bytes"] // 2),
    }


def batched_probe(sa, cases, N, B, device):
    """HBM-roofline evidence on a batch: per-kernel ms, algorithmic bytes and fraction of the
    HBM peak for B problems of cart-pole N stepped together (run AFTER the timed region)."""
    pp, sysb, _ = make_system(sa, cases, N, list(range(B)), device)
    for _ in range(3):
        sysb.reset_regularization()
        sysb.newton_step(True)
    kb, nfb, gb = kernel_groups(sysb, iters=10)
    out = {
        "workload": f"{B} x cart-pole N={N}", "batch": B, "N": N,
        "ldlt_path": "lane-per-problem interleaved" if B >= 64 else "workgroup per task",
        "steps_per_s": B / (kb["total"] * 1e-3),
        "per_kernel_ms": {k: v[0] for k, v in gb.items()},
        "algorithmic_bytes": {k: B * v[1] for k, v in gb.items()},
        "per_kernel_GBps": {k: B * v[1] / (v[0] * 1e-3) / 1e9 for k, v in gb.items()},
        "hbm_frac": {k: B * v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS for k, v in gb.items()},
        "factorizations_per_step": kb["factorizations"],
    }
    tfile = ROOT / "profiles" / f"{PROFILE_TAG}_b512xN1000_traffic.json"
    if tfile.exists() and N == 1000 and B == 512:
        tj = json.loads(tfile.read_text())
        pre = {"tape_sweep": ("tape_sweep", "slpx_tape_templates", "tape_reduce_kernel"),
               "kkt_assemble": ("kkt_assemble",), "kkt_rhs": ("kkt_rhs_kernel", "kkt_rhs_il_kernel"),
               "ldlt_factor": ("ldlt_factor_il_kernel", "il_gather_kernel", "ldlt_stats_il_kernel"),
               "ldlt_solve": ("ldlt_bwd_il_kernel",)}
        # grids with fewer than 100000 threads are the single-problem launches of the same run
        out["traffic"] = {grp: sum(e["hbm_bytes_per_launch"] for kname, grids in tj.items()
                                   if kname.startswith(pres) for grid, e in grids.items()
                                   if not (kname.startswith(("slpx_tape_templates", "tape_")) and int(grid) < 100000))
                          for grp, pres in pre.items()}
        # the same fractions on the bytes the PMC counters saw (committed pass, this run's times)
        out["hbm_frac_pmc"] = {k: out["traffic"][k] / (gb[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
                               for k in gb if out["traffic"].get(k)}
    sysb.close()
    pp.close()
    return out


def whole_solve(sa, N, seeds=9):
    """Problem::solve() at horizon N (status, iterations, wall time) — outside the timed region; the iteration path is
    the product's resident IPM (csrc/ipm.cpp) — from the benchmark's initial guess (seed 0: the reported run) AND from
    eight copies of it perturbed by 1e-13 relative (x * (1 + 1e-13 u), numpy default_rng(seed)): whether this IPM gets
    through the swing-up on a given grid hangs on the last bits of the arithmetic, in the reference algorithm itself,
    so one run per horizon says nothing about how robust its outcome is.  `robustness` puts the
    product's success fraction beside the oracle's (profiles/<tag>_oracle_robustness.json: the same experiment with the
    CPU restatement, made by profiles/oracle_robustness.py) and beside what the reference's own published sweep shows."""
    from sleipnir_amd.optimization import ExitStatus

    # (us_per_iteration = the whole interior-point iteration — step kernel, look-ahead iterate, its sweep and error
    # norms, host decisions — averaged over the solve, inertia re-attempts, backtracking, second-order corrections and
    # restoration iterations included; a first untimed solve where the horizon is short: code objects, first launches)
    runs = []
    first = None
    for k in ([-1] if N <= 500 else []) + list(range(seeds)):
        sa.lib().slpx_graph_reset()
        pp = models.cart_pole(N, 5.0 / N)
        if k > 0:
            x = pp.get_x()
            pp.set_x(x * (1 + 1e-13 * np.random.default_rng(k).uniform(-1, 1, len(x))))
        status, rep = pp.solve()
        pp.close()
        if k < 0:
            continue
        runs.append({"seed": k, "status": int(status), "iterations": int(rep["iterations"]), "t_total_s": rep["t_total"],
                     "restorations": int(rep["restorations"]), "restoration_iterations": int(rep["restoration_iterations"])})
        if k == 0:
            first = (status, rep)
    status, rep = first
    t_iter = rep["t_total"] - rep["t_restoration_setup"]
    ok = [r for r in runs if r["status"] == 0]
    robustness = {
        "perturbation": "initial guess x (1 + 1e-13 u), u ~ U(-1, 1), seeds 0 (unperturbed) .. %d" % (seeds - 1),
        "product": {"success_fraction": len(ok) / len(runs), "statuses": sorted({r["status"] for r in runs}),
                    "median_iterations_of_successes": float(np.median([r["iterations"] for r in ok])) if ok else None,
                    "median_time_s_of_successes": float(np.median([r["t_total_s"] for r in ok])) if ok else None,
                    "runs": runs},
        # reference-held evidence (benchmarks/cart-pole-scalability-results-sleipnir.csv: the harness drops
        # non-SUCCESS rows, benchmarks/scalability/util.hpp:100-106)
        "reference_published_sweep": {100: "row present (SUCCESS)", 150: "row present (SUCCESS)", 200: "row MISSING (not SUCCESS)",
                                      250: "row present (SUCCESS)", 300: "row present (SUCCESS)"}.get(N, "not in the published sweep"),
    }
    ofile = ROOT / "profiles" / f"{PROFILE_TAG}_oracle_robustness.json"
    if ofile.exists():
        o = json.loads(ofile.read_text())["horizons"].get(str(N))
        if o:
            robustness["oracle_cpu"] = {k: o[k] for k in ("success_fraction", "statuses", "median_iterations_of_successes",
                                                          "median_time_s_of_successes")}
            robustness["oracle_cpu"]["source"] = f"profiles/{PROFILE_TAG}_oracle_robustness.json (same perturbations, build container's CPU)"
    # final_error: the OUTER problem's KKT error measure at the last iterate it was evaluated on
    return {"N": N, "status": int(status), "status_name": ExitStatus(int(status)).name, "iterations": int(rep["iterations"]),
            "factorizations": int(rep["factorizations"]), "restorations": int(rep["restorations"]),
            "restoration_iterations": int(rep["restoration_iterations"]),
            "t_total_s": rep["t_total"], "t_compile_s": rep["t_compile"], "final_error": rep["final_error"],
            "us_per_iteration": 1e6 * t_iter / max(1, int(rep["iterations"])),
            "us_per_outer_iteration": 1e6 * (t_iter - rep["t_restoration"]) /
                                      max(1, int(rep["iterations"]) - int(rep["restoration_iterations"])),
            "us_per_restoration_iteration": (1e6 * rep["t_restoration"] / int(rep["restoration_iterations"])
                                             if rep["restoration_iterations"] else None),
            "factorizations_per_iteration": rep["factorizations"] / max(1, int(rep["iterations"])),
            "robustness": robustness}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed K-step blocks (at least; short blocks are repeated until they cover ~50 ms); the median is reported")
    ap.add_argument("--fixed-repeats", action="store_true", help="exactly --repeats blocks, however short")
    ap.add_argument("--workload", choices=["single", "batch512", "gfold"], default="single")
    ap.add_argument("--N", type=int, default=None, help="horizon (single: 1000, batch512: 500)")
    ap.add_argument("--batch", type=int, default=None,
                    help="single: independent replicas per GPU (1); batch512: total problems (512)")
    ap.add_argument("--per-gpu", action="store_true",
                    help="batch512: --batch problems on EVERY rank (weak scaling) instead of --batch problems "
                         "sharded over the ranks (strong scaling, BASELINE config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", "--no-batched-roofline", dest="no_batched", action="store_true",
                    help="skip the after-the-timed-region batch probes (64 x N=500, 512 x N=1000) that "
                         "carry the HBM-roofline evidence; profiling passes use this so that the "
                         "rocprofv3 averages are per-launch numbers of the timed workload")
    ap.add_argument("--no-whole-solve", action="store_true")
    ap.add_argument("--spawn-check", action="store_true",
                    help="launch self-test (runs without a GPU, gloo): every rank joins the process group, "
                         "rank 0 prints how many ranks answered and how the batch would be sharded")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(respawn_under_launcher(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")

    if args.spawn_check:
        from sleipnir_amd.dist import Comm, shard_range

        comm = Comm(backend="gloo")
        seen = comm.sum([1.0])[0]
        shards = comm.gather_rows(np.array([[comm.rank, len(shard_range(args.batch or 512, comm.rank, comm.world))]],
                                           dtype=np.float64), comm.world)
        if comm.rank == 0:
            print(json.dumps({"n_gpus": comm.world, "ranks_seen": int(seen), "pids_distinct": True,
                              "shard_sizes": [int(v) for v in shards[:, 1]]}))
        comm.close()
        return

    import torch

    import sleipnir_amd as sa
    from tests.support import cases

    # one process per GPU; RCCL ("nccl") for the barrier, the end-of-region MAX and the gather
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    comm = sa.Comm(backend="nccl")
    rank, local_rank, world = comm.rank, comm.local_rank, comm.world

    if args.workload in ("single", "gfold"):
        N = args.N or (1000 if args.workload == "single" else 100)
        B = args.batch or 1
        ids = [rank * B + b for b in range(B)]
        total_problems = world * B
        scaling = "weak"
    else:
        N = args.N or 500
        if args.per_gpu:
            # weak scaling: every rank steps its own `--batch` problems (ids continue across ranks: seed + id)
            B = args.batch or 512
            total_problems = world * B
            ids = [rank * B + b for b in range(B)]
            scaling = "weak"
        else:
            total_problems = args.batch or 512
            ids = list(sa.shard_range(total_problems, rank, world))
            B = len(ids)
            scaling = "strong"
    dt = 5.0 / N
    pp, system, setup_s = make_system(sa, cases, N, ids, local_rank, "gfold" if args.workload == "gfold" else "cart_pole")
    info = system.info
    n, me, mi = info["n"], info["m_e"], info["m_i"]

    def barrier():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    # the step loop runs on the library side (slpx_newton_steps): K steps, each one waited
    # for before the next is launched, regularization memory cleared before every step
    system.newton_steps(args.warmup)
    blocks = []
    failed = np.zeros(B, dtype=np.int32)
    # Every block times EXACTLY K steps; the median of the blocks is reported.  A block of few steps is short
    # (K = 20 at N=1000: under a millisecond, where one host hiccup is 5 %: VERDICT r04), so short blocks are
    # repeated until the blocks together cover ~50 ms (the same on every rank: decided from rank-maximum times).
    repeats = max(1, args.repeats)
    done = 0
    while done < repeats:
        barrier()
        t0 = time.perf_counter()
        info_step = system.newton_steps(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        blocks.append(float(comm.max([el])[0]))
        failed |= info_step
        barrier()
        done += 1
        if done == max(1, args.repeats) and not args.fixed_repeats:
            repeats = min(200, max(repeats, int(np.ceil(0.05 / max(1e-6, float(np.median(blocks)))))))
    assert np.all(failed == 0), "factorization failed in the timed region"
    elapsed = float(np.median(blocks))

    # per-problem results of the last step, in problem order (the only data-path-adjacent
    # collective: an all-gather of B x 4 doubles per rank)
    reg = system.regularization()
    rows = np.column_stack([np.array(ids, dtype=np.float64), failed.astype(np.float64), reg[:, 0], reg[:, 1]])
    table = comm.gather_rows(rows, total_problems)

    if rank == 0:
        kt, nf, groups = kernel_groups(system, iters=max(10, min(100, args.steps)))
        # what the inertia-correcting loop did on this state (same call sequence as one step)
        system.reset_regularization()
        system.sweep(True)
        system.assemble()
        system.rhs()
        _, reg_used, nf_check = system.compute()
        regularization = {
            "memory": "cleared before every step (each step starts like the first iteration of a solve)",
            "delta": float(reg_used[0, 0]), "gamma": float(reg_used[0, 1]), "factorizations": int(nf_check),
            "unregularized_attempt": "not launched: the symbolic phase found a structurally zero pivot"
            if info["struct_singular"] else "launched",
        }
        # A single problem's step is two launches (the AD sweep; KKT evaluation + factorization +
        # backward solve + back-substitution in one: csrc/device.hpp KktFuse / BacksubFuse,
        # ldlt_mf_step_kernel, the multifrontal step); the second one's algorithmic bytes are the SURVEY.md §8d
        # figures of the stages it performs.  `groups` (the stages as kernels of their own)
        # stays in the line as per_kernel_ms: it is what batches run and what the step falls
        # back to when the launch cannot be used.
        fused = system.time_fused_step(iters=max(10, min(100, args.steps))) if B == 1 else None
        if fused is not None:
            # (plans whose workgroups do not all fit the device at once — N=5000 — keep the
            # factorization and the solve as two launches; the figure is then their sum)
            stage_keys = ("kkt_assemble", "kkt_rhs", "ldlt_factor", "ldlt_solve")
            name = "kkt_factor_solve" if fused["one_launch"] else "kkt_factor + solve_backsub (two launches)"
            groups_step = {"tape_sweep": (fused["sweep"], groups["tape_sweep"][1]),
                           name: (fused["kkt_factor_solve"], sum(groups[k][1] for k in stage_keys) +
                                  8 * (n + 4 * mi) + 12 * info["nnz_Ai"])}
        else:
            groups_step = groups
        dom = max(groups_step, key=lambda k: groups_step[k][0] * (nf if k == "ldlt_factor" else 1.0))
        dom_ms, dom_bytes = groups_step[dom]
        achieved = B * dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per step of each kernel group from the committed PMC passes
        # (profiles/collect.py; FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE)
        traffic_by_group = None
        tfile = ROOT / "profiles" / f"{PROFILE_TAG}_traffic.json"
        if tfile.exists() and args.workload == "single" and N == 1000 and B == 1:
            tj = json.loads(tfile.read_text())
            prefix = {"tape_sweep": ("tape_sweep", "slpx_tape_templates"),
                      "kkt_factor_solve": ("ldlt_mf_step_kernel",),
                      "kkt_assemble": "kkt_assemble_kernel",
                      "kkt_rhs": "kkt_rhs_kernel", "ldlt_factor": "ldlt_factor_kernel",
                      "ldlt_solve": ("ldlt_fwd", "ldlt_bwd", "ldlt_mf_solve_kernel")}
            traffic_by_group = {}
            for grp, pre in prefix.items():
                pres = pre if isinstance(pre, tuple) else (pre,)
                names = [k for k in tj if k.startswith(pres)]
                # (the step kernel has two variants in the same run: the one of a chained step — last template
                # argument true, waits for its sweep — and the one the launch-duration probe times: that one)
                if sum(k.startswith("ldlt_mf_step_kernel") for k in names) > 1:
                    names = [k for k in names if not k.startswith("ldlt_mf_step_kernel") or k.endswith("false>")]
                # (a kernel launched with two grid sizes in the profiled run — the step kernel with and without a
                # workgroup for the sweep's separable sums — counts once: the grid launched most often)
                traffic_by_group[grp] = sum(max(tj[k].values(), key=lambda e: e.get("launches_sampled", 0))["hbm_bytes_per_launch"]
                                            for k in names)
        single = args.workload in ("single", "gfold") and B == 1
        roofline = {
            "bound": "hbm", "kernel": dom,
            "kernel_symbol": ("ldlt_mf_step_kernel<.., false>"
                              if fused is not None and dom.startswith("kkt_factor") and fused["one_launch"] else None),
            "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None if traffic_by_group is None else traffic_by_group.get(dom),
            "traffic_per_kernel": traffic_by_group,
            "algorithmic_bytes_per_launch": B * dom_bytes, "launch_ms": dom_ms,
            "step_launches_ms": {k: v[0] for k, v in groups_step.items()},
            "step_launches_algorithmic_bytes": {k: B * v[1] for k, v in groups_step.items()},
            "per_kernel_ms": {k: v[0] for k, v in groups.items()},
            "per_kernel_GBps": {k: B * v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else None
                                for k, v in groups.items()},
            "per_kernel_hbm_frac": {k: B * v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if v[0] > 0 else None
                                    for k, v in groups.items()},
            "factorizations_per_step": kt["factorizations"],
            "regularization": regularization,
            "note": (f"single N={N} problem: every kernel is dependency-latency bound (SURVEY.md §7 hard "
                     "part 1); the HBM fractions that mean something are in `batched` (same kernels' "
                     "batch variants on 64 x N=500 and 512 x N=1000).  launch_ms: back-to-back launches of "
                     "the step kernel on its own (variant <.., false>: profiles/*_kernel_stats.csv has it "
                     "as a row of its own); in the timed loop consecutive steps are chained — the variant "
                     "<.., true> is dispatched beside the sweep, and its row's span includes waiting for it") if single else
                    f"{B} problems of N={N} per launch on this rank",
        }
        workload = ("%s N=%d, %d problem(s) per GPU (replicas), seeded interior "
                    "IPM state, inputs resident in HBM" % ("cart-pole direct transcription" if args.workload == "single"
                                                           else "g-fold powered-descent OCP", N, B)) if args.workload != "batch512" else (
                    "batch of %d independent cart-pole N=%d problems (seed + b), sharded contiguously: %d on "
                    "rank 0, inputs resident in HBM" % (total_problems, N, B))
        out = {
            "metric": "Newton steps/sec, %s N=%d" % ("g-fold OCP" if args.workload == "gfold" else
                                                     "cart-pole direct-transcription", N),
            "value": total_problems * args.steps / elapsed,
            "unit": "Newton steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "timing": {"repeats": len(blocks), "statistic": "median of the K-step blocks",
                       "block_ms": [1e3 * b for b in blocks]},
            "config": {"workload": workload, "problems_total": total_problems,
                       "n": n, "m_e": me, "m_i": mi, "nnz_lhs": info["nnz_lhs"],
                       "nnz_L": info["nnz_L"], "etree_height": info["etree_height"],
                       "ldlt_levels": info.get("ldlt_levels"), "ldlt_supernodes": info.get("ldlt_supernodes"),
                       "ldlt_rounds": info["ldlt_rounds"], "ldlt_tasks": info["ldlt_tasks"],
                       "tape_tasks": info["tape_tasks"], "tape_nodes": info["tape_nodes"],
                       "tape_slots": info["tape_slots"],
                       "multi_gpu": "replicas only" if args.workload != "batch512" else
                                    ("%d problems on every rank, no data-path collective" % B if args.per_gpu else
                                     "problems sharded, no data-path collective")},
            "ms_per_ldlt_factor": groups["ldlt_factor"][0],
            "ms_per_ldlt_solve": groups["ldlt_solve"][0],
            "setup_s": setup_s,
            "roofline": roofline,
        }
        if args.workload == "batch512":
            out["per_problem"] = {
                "columns": ["problem", "status", "delta", "gamma"],
                "failed": int(np.sum(table[:, 1] != 0)),
                "delta_histogram": {f"{d:.0e}": int(c) for d, c in zip(*np.unique(table[:, 2], return_counts=True))},
                "gamma_histogram": {f"{g:.0e}": int(c) for g, c in zip(*np.unique(table[:, 3], return_counts=True))},
                "first_rows": table[:8].tolist(),
                # the whole gathered table, for comparing a sharded run with a single-process one
                # (tests/test_multi_gpu_readiness_gpu.py): problem order, status, delta, gamma
                "table_sha256": __import__("hashlib").sha256(np.ascontiguousarray(table).tobytes()).hexdigest(),
                "rows": int(table.shape[0]),
            }
    system.close()
    pp.close()
    if rank == 0:
        if world == 1 and single and args.workload == "single":
            # warm setup: the same model compiled a second time in this process (hipRTC code
            # objects cached, HIP runtime up)
            pp2, sys2, setup2 = make_system(sa, cases, N, [0], local_rank)
            out["setup_s"] = {"model": setup_s["model"], "compile_and_upload": setup_s["compile_and_upload"],
                              "compile_and_upload_warm": setup2["compile_and_upload"]}
            sys2.close()
            pp2.close()
            if not args.no_batched:
                out["batched"] = [batched_probe(sa, cases, 500, 64, local_rank),
                                  batched_probe(sa, cases, 1000, 512, local_rank)]
                # BASELINE config 4 (512 x N=500 over 8 GPUs) before a node measures it: the per-GPU share (64
                # problems) and the whole batch on this one GPU, both MEASURED here; their ratio x 8 is a
                # PROJECTION of the 8-GPU run (no collective on the data path: the shards are independent)
                share = out["batched"][0]["steps_per_s"]
                whole = batched_probe(sa, cases, 500, 512, local_rank)["steps_per_s"]
                out["config4_projection"] = {
                    "per_gpu_share_64xN500_steps_per_s": share, "one_gpu_512xN500_steps_per_s": whole,
                    "projected_8_gpu_steps_per_s": 8.0 * share, "projected_speedup_on_8_gpus": 8.0 * share / whole,
                    "kind": "projection from two single-GPU measurements, not an 8-GPU measurement"}
            if not args.no_whole_solve:
                # Problem::solve() at the BASELINE horizon and at three shorter ones.  Whether this
                # IPM gets through the swing-up on a given grid depends on the last bits of the
                # arithmetic — in the reference algorithm itself (DESIGN.md §2,
                # profiles/r02_oracle_sensitivity.txt: the CPU restatement ends LOCALLY_INFEASIBLE
                # at N=1000 as well) — so the line shows more than one horizon.
                # (N=100/150/250/300 succeed and N=200 does not in the reference's published sweep)
                out["whole_solve"] = whole_solve(sa, N)
                out["whole_solves"] = [whole_solve(sa, n_) for n_ in (100, 150, 200, 250, 300, 500)] + [out["whole_solve"]]
        if not args.no_cpu_baseline and world == 1 and args.workload != "gfold":
            out["cpu_baseline"] = cpu_baseline(N, dt)
            out["speedup_vs_cpu_baseline"] = out["value"] / (out["cpu_baseline"]["value"] * 1.0)
        print(json.dumps(out))
    comm.close()


if __name__ == "__main__":
    main()
