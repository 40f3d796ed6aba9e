"""The N > 1 path on CPU: world_size-2 gloo.  A batch of independent problems is sharded
contiguously across ranks (sleipnir_amd.dist.shard_range), every rank steps its own
problems with NO data-path collective, and the end-of-region collectives are exactly the
ones bench.py uses on RCCL: MAX of the elapsed time, SUM of counters, a gather of
per-problem result rows.  The Newton steps themselves are produced by the host
interpreter of the compiled plans (tests/support/hostcheck — test infrastructure; on
the GPU box the same ranks call the HIP kernels)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
from tests.support import models

ROOT = Path(__file__).resolve().parents[1]


def test_shard_range_partitions():
    from sleipnir_amd.dist import shard_range

    for n_items in (0, 1, 5, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(shard_range(n_items, r, world))
            assert seen == list(range(n_items))
            sizes = [len(shard_range(n_items, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert len(shard_range(512, 3, 8)) == 64  # config 4: 512 problems over 8 GPUs
    # the C-ABI's rule (what a C++ host uses, include/slpx.h: slpx_shard_range) is the same one
    import ctypes

    import sleipnir_amd as sa

    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for n_items in (0, 5, 512, 513):
        for world in (1, 3, 8):
            for r in range(world):
                assert sa.lib().slpx_shard_range(n_items, r, world, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert range(lo.value, hi.value) == shard_range(n_items, r, world)
    assert sa.lib().slpx_shard_range(4, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == -1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _step_rows(problem_ids, N):
    """One Newton step per problem id (seeded interior state); returns rows
    [id, ‖p‖₁, ‖p_s‖₁, ‖p_z‖₁, n_pos, n_neg]."""
    import sleipnir_amd as sa
    from tests.support import cases, hostcheck

    sa.lib().slpx_graph_reset()
    pp = models.cart_pole(N, 5.0 / N)
    hc = hostcheck.HostCheck(pp)
    n, me, mi = hc.n, hc.m_e, hc.m_i
    x0 = pp.get_x()
    rows = []
    for b in problem_ids:
        x, s, y, z, mu = cases.newton_state("interior", x0, n, me, mi, 1.0, seed=cases.SEED + b)
        hc.sweep(x, y, z, True)
        hc.assemble(s, z)
        hc.rhs(s, y, z, mu)
        D, stats = hc.factor(1e-4, 1e-10)
        p = hc.solve()
        ps, pz = hc.backsub(s, z, mu)
        rows.append([b, np.abs(p).sum(), np.abs(ps).sum(), np.abs(pz).sum(), stats[0], stats[1]])
    hc.close()
    pp.close()
    return np.array(rows, dtype=np.float64).reshape(len(rows), 6)


def _worker(rank, world, port, n_problems, N, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from sleipnir_amd.dist import Comm, shard_range

    comm = Comm(backend="gloo")
    mine = shard_range(n_problems, comm.rank, comm.world)
    comm.barrier()
    rows = _step_rows(list(mine), N)
    elapsed = comm.max([1.0 + comm.rank])          # MAX over ranks
    total = comm.sum([float(len(mine))])           # SUM of counters
    table = comm.gather_rows(rows, n_problems)     # per-problem results in problem order
    comm.barrier()
    if comm.rank == 0:
        np.savez(Path(out_dir) / "result.npz", elapsed=elapsed, total=total, table=table)
    comm.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharded_steps(tmp_path, slpx, hostcheck):
    import torch.multiprocessing as mp

    n_problems, N, world = 5, 6, 2
    mp.spawn(_worker, args=(world, _free_port(), n_problems, N, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "result.npz")
    assert res["elapsed"][0] == 2.0                 # max(1 + rank)
    assert res["total"][0] == n_problems
    want = _step_rows(list(range(n_problems)), N)   # single-process reference
    assert res["table"].shape == want.shape
    assert np.array_equal(res["table"][:, 0], np.arange(n_problems))
    assert np.allclose(res["table"], want, rtol=1e-12, atol=0)


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` outside a launcher must become two ranks (the flag
    used to be parsed and ignored).  --spawn-check stops after the process group is up, so it runs
    without a GPU (gloo)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--spawn-check", "--batch", "513"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["shard_sizes"] == [257, 256]
    # under a launcher the flag must agree with the world size instead of being ignored
    env["WORLD_SIZE"] = "4"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--spawn-check"],
                         capture_output=True, text=True, env=env, timeout=60)
    assert res.returncode != 0 and "WORLD_SIZE=4" in res.stderr
