"""One-process-per-GPU data parallelism for the separation path.

Mixtures are independent (no op of the model mixes batch entries: LayerNorm is per token, there is no
BatchNorm on the path -- SURVEY.md 8e), so inference shards by batch and needs NO data-path collective.
RCCL (torch.distributed backend "nccl" on ROCm) is used only for: the timing barrier, MAX of the elapsed
time, and the SUM of per-rank metric accumulators at the end of an evaluation.  The same code runs on
CPU with the gloo backend (tests/test_parallel_gloo.py).

Semantic note restated from the reference: PIT in loss/metric is BATCH level (ss_losses.py:21-26,
base_metric.py:41-60) -- each rank picks the permutation on its own shard, which equals running the reference
with that shard as its batch, not one global-batch step.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n_items for `rank`; sizes differ by at most one, earlier ranks larger."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class DistEnv:
    rank: int
    local_rank: int
    world: int
    device: torch.device
    backend: Optional[str]
    active: bool = False      # a process group exists: collectives go through torch.distributed (always when world > 1)

    @classmethod
    def from_environ(cls, expected_world: Optional[int] = None, backend: Optional[str] = None,
                     device: Optional[str] = None, force_init: bool = False) -> "DistEnv":
        """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).
        `force_init`: create the process group even for ONE rank, so that every collective of this class and the
        gradient all-reduce really go through the backend (RCCL on a one-GPU box: tests/test_gpu_parallel.py)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if expected_world is not None and expected_world != world:
            raise RuntimeError(f"--gpus {expected_world} but WORLD_SIZE={world}: launch with "
                               f"`python -m torch.distributed.run --nproc-per-node {expected_world} ...`")
        if device is None:
            device = f"cuda:{local}"
        dev = torch.device(device)
        active = world > 1 or force_init
        if active:
            import torch.distributed as dist
            backend = backend or ("nccl" if dev.type == "cuda" else "gloo")   # "nccl" IS RCCL on ROCm
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                kw = {"device_id": dev} if backend == "nccl" else {}
                dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        else:
            backend = None
        return cls(rank, local, world, dev, backend, active)

    # -- collectives (all trivially local when world == 1) -------------------------------------------
    def barrier(self):
        if self.active:
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.barrier(device_ids=[self.device.index])
            else:
                dist.barrier()

    def _reduce(self, values: Sequence[float], op) -> List[float]:
        if not self.active:
            return list(values)
        import torch.distributed as dist
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return t.cpu().tolist()

    def max_over_ranks(self, x: float) -> float:
        import torch.distributed as dist
        return self._reduce([x], dist.ReduceOp.MAX)[0]

    def sum_over_ranks(self, values: Sequence[float]) -> List[float]:
        import torch.distributed as dist
        return self._reduce(values, dist.ReduceOp.SUM)

    def gather_over_ranks(self, x: float) -> List[float]:
        """[x of rank 0, x of rank 1, ...] on every rank (one all_gather of a float64 scalar)."""
        if not self.active:
            return [float(x)]
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [float(o.cpu()[0]) for o in out]

    def backend_world(self) -> int:
        """World size as the BACKEND sees it (dist.get_world_size()), 0 without a process group."""
        if not self.active:
            return 0
        import torch.distributed as dist
        return int(dist.get_world_size())

    def close(self):
        if self.active:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
