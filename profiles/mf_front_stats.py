#!/usr/bin/env python3
"""The fronts of the multifrontal plan level by level (host only: the plan, no device): how many fronts a level of a
task has (16 waves take one pass of sixteen), how wide they are, how many children values an entry sums — what the
levers of the step kernel's level time would act on (children pre-summed, half-wave fronts).

    PYTHONPATH=$PWD python profiles/mf_front_stats.py [N]
"""
import os
import sys

os.environ.setdefault("SLPX_LDLT_MF", "1")  # (the host handle builds the fronts when asked)
from collections import Counter

import numpy as np

from tests.support import hostcheck, models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pp = models.cart_pole(N, 5.0 / N)
h = hostcheck.HostCheck(pp)
F = h.mf_fronts()
print("plan", h.mf_plan())
tasks = np.unique(F[:, 0])
by_round = {}
for t in tasks:
    ft = F[F[:, 0] == t]
    by_round.setdefault(int(ft[0, 1]), []).append(ft)
for r in sorted(by_round):
    fts = by_round[r]
    nl = Counter(int(ft[:, 2].max()) + 1 for ft in fts)
    print(f"round {r}: {len(fts)} tasks, levels per task {dict(nl)}")
    # the task with the most levels, then the most fronts: level by level
    ft = max(fts, key=lambda a: (a[:, 2].max(), len(a)))
    for l in range(int(ft[:, 2].max()) + 1):
        fl = ft[ft[:, 2] == l]
        desc = Counter((int(a[3]), int(a[4]), int(a[5]), int(a[6])) for a in fl)
        shown = ", ".join(f"{c}x(w{w} nr{nr} nch{nch} ns{ns})" for (w, nr, nch, ns), c in sorted(desc.items(), key=lambda kv: -kv[1])[:8])
        print(f"   task {int(ft[0, 0])} level {l}: {len(fl)} fronts; {shown}")
    # over all tasks of the round: per level, the distribution of the front count and of the worst (w, nch)
    L = max(int(a[:, 2].max()) + 1 for a in fts)
    for l in range(L):
        counts, worst = [], Counter()
        for a in fts:
            fl = a[a[:, 2] == l]
            if len(fl) == 0:
                continue
            counts.append(len(fl))
            k = max(fl, key=lambda q: (q[3] * (1 + q[5]) + q[6] / 16.0))
            worst[(int(k[3]), int(k[4]), int(k[5]), int(k[6]))] += 1
        print(f"   level {l}: tasks {len(counts)} fronts min/median/max {min(counts)}/{int(np.median(counts))}/{max(counts)}; heaviest front (w, nr, nch, n_s): {worst.most_common(4)}")
# passes of sixteen waves per level, task by task (a level of 17 fronts costs two fronts' time)
for r in sorted(by_round):
    hist = Counter()
    for a in by_round[r]:
        passes = tuple(int((np.sum(a[:, 2] == l) + 15) // 16) for l in range(int(a[:, 2].max()) + 1))
        hist[passes] += 1
    print(f"round {r}: passes per level -> tasks: {dict(hist)}")
