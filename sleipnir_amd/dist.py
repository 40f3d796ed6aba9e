"""Multi-GPU plumbing: one process per GPU, problems sharded across
ranks, NO exchange inside a Newton step.  The only collectives are end-of-region
reductions (MAX of elapsed time, SUM of counters) and a gather of per-problem results —
RCCL (`backend="nccl"`) on GPUs, gloo in the CPU tests.  torch.distributed is plumbing;
nothing here computes.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of a batch of independent problems owned by `rank`.  The first
    `n_items % world` ranks take one extra item (multistart.hpp:52-62 hands whole
    solves to threads the same way)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


class Comm:
    """Thin wrapper over torch.distributed that degrades to a no-op at world size 1."""

    def __init__(self, backend: str | None = None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kwargs = {}
            if backend == "nccl":
                self.device = torch.device("cuda", self.local_rank)
                kwargs["device_id"] = self.device
            else:
                self.device = torch.device("cpu")
            dist.init_process_group(backend=backend, **kwargs)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, values, op_name):
        values = np.atleast_1d(np.asarray(values, dtype=np.float64))
        if self.dist is None:
            return values
        import torch

        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op_name))
        return t.cpu().numpy()

    def max(self, values):
        return self._reduce(values, "MAX")

    def sum(self, values):
        return self._reduce(values, "SUM")

    def gather_rows(self, rows: np.ndarray, n_total: int) -> np.ndarray:
        """All-gather of per-problem result rows ([n_local, k] on each rank, sharded
        with shard_range) into [n_total, k] in problem order."""
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if self.dist is None:
            return rows
        import torch

        k = rows.shape[1]
        cap = len(shard_range(n_total, 0, self.world))  # largest shard
        pad = np.zeros((cap, k))
        pad[: rows.shape[0]] = rows
        mine = torch.tensor(pad, dtype=torch.float64, device=self.device)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        parts = [o.cpu().numpy()[: len(shard_range(n_total, r, self.world))] for r, o in enumerate(out)]
        return np.concatenate(parts, axis=0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
