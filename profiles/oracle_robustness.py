#!/usr/bin/env python3
"""Whole solves of the cart-pole swing-up by the ORACLE (CPU restatement of the reference's interior_point.hpp) from the
benchmark's initial guess and from eight copies of it perturbed by 1e-13 relative (x * (1 + 1e-13 u), u uniform in
[-1, 1], numpy default_rng(seed), seed 0 = unperturbed) — the experiment bench.py makes with the product on the GPU
(`whole_solves[*].robustness`), on the algorithm-faithful CPU side.  Output: profiles/<tag>_oracle_robustness.json, which
bench.py puts beside the product's numbers.  CPU only, minutes:
    PYTHONPATH=$PWD python profiles/oracle_robustness.py r06 [N ...]"""
import json
import sys
import time

import numpy as np

from tests.support import oracle

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
horizons = [int(a) for a in sys.argv[2:]] or [100, 150, 200, 250, 300, 500, 1000]
out = {"what": "oracle (CPU restatement of the reference IPM), cart-pole swing-up, initial guess x (1 + 1e-13 u), 9 seeds (0 = unperturbed)",
       "horizons": {}}
for N in horizons:
    runs = []
    for k in range(9):
        oracle.lib().orc_reset()
        op = oracle.OracleProblem.cart_pole(N, 5.0 / N)
        x = op.get_x()
        if k:
            x = x * (1 + 1e-13 * np.random.default_rng(k).uniform(-1, 1, len(x)))
        op.set_x(x)
        t0 = time.perf_counter()
        st, stats = op.solve()
        runs.append({"seed": k, "status": int(st), "iterations": int(stats["iterations"]), "t_s": time.perf_counter() - t0})
        print(N, runs[-1], flush=True)
    ok = [r for r in runs if r["status"] == 0]
    out["horizons"][str(N)] = {
        "runs": runs, "success_fraction": len(ok) / len(runs),
        "median_iterations_of_successes": float(np.median([r["iterations"] for r in ok])) if ok else None,
        "median_time_s_of_successes": float(np.median([r["t_s"] for r in ok])) if ok else None,
        "statuses": sorted({r["status"] for r in runs}),
    }
    json.dump(out, open(f"profiles/{tag}_oracle_robustness.json", "w"), indent=1)
