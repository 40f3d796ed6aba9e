#!/usr/bin/env python3
"""bench.py's batched probe on its own: per-kernel ms and HBM fractions of a batch stepped together.
    PYTHONPATH=$PWD python profiles/batched_probe.py [N] [batch]"""
import json
import sys

import bench
import sleipnir_amd as sa
from tests.support import cases

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
out = bench.batched_probe(sa, cases, N, B, 0)
print(json.dumps({k: out[k] for k in ("workload", "steps_per_s", "per_kernel_ms", "hbm_frac", "factorizations_per_step")}))
