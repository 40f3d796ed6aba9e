"""GPU tier, VERDICT r03 item 9: the multi-GPU path run end to end at world size = the number of
devices on the box — 1 on the driver's test box, N on a node — so that the first 8-GPU lease is a
measurement, not a debugging session.

  * bench.py under the DRIVER's launcher command (python -m torch.distributed.run --nnodes=1
    --nproc-per-node W --master-addr 127.0.0.1 ... bench.py --gpus W --workload batch512): one process
    per GPU over RCCL, the batch sharded (config 4, strong scaling) and the weak-scaling mode
    (--per-gpu); the gathered per-problem table (problem, status, delta, gamma) must be the table of ONE
    process stepping all the problems — same order, same regularization, nothing failed;
  * the C++ host of INTEGRATION.md §5 (RCCL barrier / MAX / all-gather) with W ranks against its own
    single-process run, bit for bit."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
FAST = ["--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-batched", "--no-whole-solve"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(world, extra, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [str(ROOT / "bench.py"), "--gpus", str(world), "--workload", "batch512", "--N", "100", *extra, *FAST]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_sharded_and_weak_scaling_tables_equal_the_single_process_ones(slpx):
    world = slpx.lib().slpx_device_count()
    assert world >= 1
    per_gpu = 64
    total = per_gpu * world
    # config 4's mode: `total` problems sharded over the ranks, under the driver's launcher command
    sharded = _bench(world, ["--batch", str(total)], launcher=True)
    assert sharded["n_gpus"] == world and sharded["scaling"] == "strong" and sharded["value"] > 0
    assert sharded["config"]["problems_total"] == total
    # weak scaling: every rank its own 64 problems (ids continue across the ranks)
    weak = _bench(world, ["--batch", str(per_gpu), "--per-gpu"], launcher=True)
    assert weak["n_gpus"] == world and weak["scaling"] == "weak" and weak["config"]["problems_total"] == total
    # ONE process, one GPU, all the problems
    whole = _bench(1, ["--batch", str(total)], launcher=False)
    for line in (sharded, weak):
        pp, pw = line["per_problem"], whole["per_problem"]
        assert pp["rows"] == total and pp["failed"] == 0
        assert pp["table_sha256"] == pw["table_sha256"], (pp["first_rows"], pw["first_rows"])
    print(f"world {world}: sharded {sharded['value']:.0f}, weak {weak['value']:.0f}, one process {whole['value']:.0f} steps/s")


def test_cxx_rccl_host_at_device_count_gives_the_rows_of_one_process(slpx, tmp_path):
    from tests import test_multi_gpu_host as host

    world = slpx.lib().slpx_device_count()
    host.build(slpx)
    total, N, steps = 64 * world, 100, 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SLPX_NCCL_ID_FILE=str(tmp_path / "id"), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([str(host.BIN), str(total), str(N), str(steps)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == world and line["rows"] == total and line["rows_out_of_order_or_failed"] == 0
    # the same problems in ONE process without RCCL: the same (delta, gamma) for every problem, in order
    whole = host.run([total, N, steps, "--no-comm"], RANK=0, WORLD_SIZE=1)
    assert whole.returncode == 0, whole.stdout + whole.stderr
    one = json.loads([l for l in whole.stdout.splitlines() if l.startswith("{")][-1])
    assert one["rows"] == total and one["regularization_hash"] == line["regularization_hash"], (one, line)
