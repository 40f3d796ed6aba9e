#!/usr/bin/env python3
"""Newton steps/s of the interior-point Newton step on the cart-pole direct-
transcription problem (BASELINE.json: N=1000, single problem per GPU, fp64).

One "step" = AD refresh {g, A_e, H} + KKT lhs/rhs assembly + regularized LDLᵀ
(inertia-correcting loop, every attempt a device factorization) + solve + (p_s, p_z)
back-substitution, i.e. interior_point.hpp:809-812 + :426-482, on inputs already
resident in HBM (the seeded interior state of SURVEY.md §8d).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by torch.distributed.run (one rank per GPU): every rank steps its
own replica(s) of the problem — the path has no cross-problem exchange (SURVEY.md
§8e) — and the only collective is the MAX of the elapsed time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--N", type=int, default=1000, help="horizon (BASELINE config: 1000)")
    ap.add_argument("--batch", type=int, default=1, help="independent problems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-roofline", action="store_true",
                    help="also step a batch of independent problems (the configuration on which "
                         "the HBM roofline of the gather kernels is measurable) and add a "
                         "'batched' object; off by default so that the default command launches "
                         "one kernel shape only and the rocprofv3 --stats averages under "
                         "profiles/ are per-launch numbers of the N=1 workload")
    ap.add_argument("--no-batched-roofline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--roofline-batch", type=int, default=512)
    args = ap.parse_args()

    import torch

    import sleipnir_amd as sa
    from tests.support import cases

    # one process per GPU; RCCL ("nccl") only for the end-of-region MAX
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    comm = sa.Comm(backend="nccl")
    rank, local_rank, world = comm.rank, comm.local_rank, comm.world

    N = args.N
    dt = 5.0 / N
    sa.lib().slpx_graph_reset()
    t0 = time.perf_counter()
    pp = sa.Problem.cart_pole(N, dt)
    t_model = time.perf_counter() - t0
    t0 = time.perf_counter()
    system = sa.System(pp, batch=args.batch, device=local_rank)
    t_compile = time.perf_counter() - t0
    info = system.info
    n, me, mi = info["n"], info["m_e"], info["m_i"]

    # scaling at x0 (problem_scaling.hpp:100-107) from an unscaled device sweep
    x0 = pp.get_x()
    B = args.batch
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
    system.set_scaling(scales)

    # device-resident inputs: seeded interior states, one per problem
    states = [cases.newton_state("interior", x0, n, me, mi, scales[0], seed=cases.SEED + rank * B + b)
              for b in range(B)]
    system.set_state(np.stack([s_[0] for s_ in states]), np.stack([s_[1] for s_ in states]),
                     np.stack([s_[2] for s_ in states]), np.stack([s_[3] for s_ in states]),
                     np.array([s_[4] for s_ in states]))

    def barrier():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    nfact_total = 0
    # the step loop runs on the library side (slpx_newton_steps): K steps, each one waited
    # for before the next is launched, regularization memory cleared before every step
    system.newton_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    info_step = system.newton_steps(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert np.all(info_step == 0), "factorization failed in the timed region"
    elapsed = float(comm.max([elapsed])[0])
    barrier()

    if rank == 0:
        # per-kernel-group durations measured live with HIP events on the library's stream
        system.reset_regularization()
        kt = system.time_step(iters=max(10, min(100, args.steps)), refresh_ad=True)
        nf = max(1.0, kt["factorizations"])
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
        groups = {
            "tape_sweep": (kt["sweep"], info["sweep_bytes"]),
            "kkt_assemble": (kt["assemble"], info["assemble_bytes"]),
            "kkt_rhs": (kt["rhs"], info["rhs_bytes"]),
            "ldlt_factor": (kt["factor"] / nf, info["factor_bytes"]),
            # the timed launch is the BACKWARD substitution only (the forward one rides in the
            # factorization): half of the 32 l + 16(n + m_e) of SURVEY.md §8d
            "ldlt_solve": (kt["solve"], info["solve_bytes"] // 2),
        }
        # the tape program (static, read once per sweep) belongs to the sweep's bytes just
        # like the index maps belong to kkt_assemble's
        groups["tape_sweep"] = (kt["sweep"], info["sweep_bytes"] + info["tape_program_bytes"])
        dom = max(groups, key=lambda k: groups[k][0] * (nf if k == "ldlt_factor" else 1.0))
        dom_ms, dom_bytes = groups[dom]
        achieved = B * dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per step of each kernel group from the committed PMC passes
        # (profiles/collect.py; FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE)
        traffic_by_group = None
        tfile = ROOT / "profiles" / "r01_traffic.json"
        if tfile.exists() and N == 1000 and B == 1:
            tj = json.loads(tfile.read_text())
            prefix = {"tape_sweep": ("tape_sweep", "slpx_tape_templates", "tape_reduce_kernel"),
                      "kkt_assemble": "kkt_assemble_kernel",
                      "kkt_rhs": "kkt_rhs_kernel", "ldlt_factor": "ldlt_factor_kernel",
                      "ldlt_solve": ("ldlt_fwd_kernel", "ldlt_bwd_kernel")}
            traffic_by_group = {}
            for grp, pre in prefix.items():
                pres = pre if isinstance(pre, tuple) else (pre,)
                traffic_by_group[grp] = sum(e["hbm_bytes_per_launch"] for kname, grids in tj.items()
                                            if kname.startswith(pres) for e in grids.values())
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None if traffic_by_group is None else traffic_by_group[dom],
            "traffic_per_kernel": traffic_by_group,
            "algorithmic_bytes_per_launch": B * dom_bytes, "launch_ms": dom_ms,
            "per_kernel_ms": {k: v[0] for k, v in groups.items()},
            "per_kernel_GBps": {k: B * v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else None
                                for k, v in groups.items()},
            "factorizations_per_step": kt["factorizations"],
            "regularization": regularization,
            "note": "single N=1000 problem: every kernel is dependency-latency bound (SURVEY.md "
                    "§7 hard part 1); HBM fractions are meaningful on a batch (--batched-roofline; "
                    "DESIGN.md §4, profiles/r01_batched_*)",
        }
        out = {
            "metric": "Newton steps/sec, cart-pole direct-transcription N=%d" % N,
            "value": world * B * args.steps / elapsed,
            "unit": "Newton steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cart-pole direct transcription N=%d, %d problem(s) per GPU, "
                                   "seeded interior IPM state, inputs resident in HBM" % (N, B),
                       "n": n, "m_e": me, "m_i": mi, "nnz_lhs": info["nnz_lhs"],
                       "nnz_L": info["nnz_L"], "etree_height": info["etree_height"],
                       "ldlt_rounds": info["ldlt_rounds"], "ldlt_tasks": info["ldlt_tasks"],
                       "tape_tasks": info["tape_tasks"], "tape_nodes": info["tape_nodes"],
                       "tape_slots": info["tape_slots"], "multi_gpu": "replicas only"},
            "ms_per_ldlt_factor": groups["ldlt_factor"][0],
            "ms_per_ldlt_solve": groups["ldlt_solve"][0],
            "setup_s": {"model": t_model, "compile_and_upload": t_compile},
            "roofline": roofline,
        }
        if args.batched_roofline and world == 1 and B == 1:
            # the configuration on which HBM-roofline claims are measurable
            RB = args.roofline_batch
            sysb = sa.System(pp, batch=RB, device=local_rank)
            sysb.set_scaling(scales)
            st = [cases.newton_state("interior", x0, n, me, mi, scales[0], seed=cases.SEED + b)
                  for b in range(RB)]
            sysb.set_state(np.stack([s_[0] for s_ in st]), np.stack([s_[1] for s_ in st]),
                           np.stack([s_[2] for s_ in st]), np.stack([s_[3] for s_ in st]),
                           np.array([s_[4] for s_ in st]))
            for _ in range(3):
                sysb.reset_regularization()
                sysb.newton_step(True)
            sysb.reset_regularization()
            kb = sysb.time_step(iters=10, refresh_ad=True)
            nfb = max(1.0, kb["factorizations"])
            gb = {"tape_sweep": (kb["sweep"], info["sweep_bytes"]),
                  "kkt_assemble": (kb["assemble"], info["assemble_bytes"]),
                  "kkt_rhs": (kb["rhs"], info["rhs_bytes"]),
                  "ldlt_factor": (kb["factor"] / nfb, info["factor_bytes"]),
                  "ldlt_solve": (kb["solve"], info["solve_bytes"] // 2)}
            # HBM bytes per launch group from the committed PMC passes of this configuration
            btraffic = None
            bfile = ROOT / "profiles" / "r01_batched_traffic.json"
            if bfile.exists() and N == 1000 and RB == 512:
                tj = json.loads(bfile.read_text())
                pre = {"tape_sweep": ("tape_sweep", "slpx_tape_templates", "tape_reduce_kernel"),
                       "kkt_assemble": ("kkt_assemble",), "kkt_rhs": ("kkt_rhs_kernel",),
                       "ldlt_factor": ("ldlt_factor_il_kernel", "il_gather_kernel", "ldlt_stats_il_kernel"),
                       "ldlt_solve": ("ldlt_bwd_il_kernel",)}
                # grids with fewer than 1000 workgroup-threads are the single-problem launches of the same run
                btraffic = {grp: sum(e["hbm_bytes_per_launch"] for kname, grids in tj.items()
                                     if kname.startswith(pres) for grid, e in grids.items()
                                     if not (kname.startswith(("slpx_tape_templates", "tape_")) and int(grid) < 100000))
                            for grp, pres in pre.items()}
            out["batched"] = {
                "batch": RB, "steps_per_s": RB / (kb["total"] * 1e-3),
                "per_kernel_ms": {k: v[0] for k, v in gb.items()},
                "per_kernel_GBps": {k: RB * v[1] / (v[0] * 1e-3) / 1e9 for k, v in gb.items()},
                "per_kernel_hbm_frac": {k: RB * v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                        for k, v in gb.items()},
                "factorizations_per_step": kb["factorizations"],
                "algorithmic_bytes": {k: RB * v[1] for k, v in gb.items()},
                "traffic": btraffic,
            }
            sysb.close()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, dt)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    system.close()
    comm.close()


if __name__ == "__main__":
    main()
