"""Data-parallel evaluation loop: the per-batch semantics of the reference's Inferencer.process_batch
(src/trainer/inferencer.py:98-127) and MetricTracker (src/metrics/tracker.py:29-42: running mean of per-batch
values), sharded over ranks by speech_separation_amd.parallel with one SUM all-reduce at the end."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Mapping, Optional

import torch

from .parallel import DistEnv, shard_range


def move_batch_to_device(batch: Mapping[str, object], device, device_tensors: Iterable[str]) -> Dict[str, object]:
    """base_trainer.py:343-356: every name in device_tensors must be present and is moved."""
    out = dict(batch)
    for name in device_tensors:
        out[name] = out[name].to(device)
    return out


def evaluate(model: Callable[..., Mapping[str, torch.Tensor]], batches: List[Mapping[str, object]], metrics: List[object],
             env: Optional[DistEnv] = None, device_tensors: Iterable[str] = ("mix", "s1", "s2", "s1_embedding",
                                                                             "s2_embedding")) -> Dict[str, float]:
    """Each rank processes its contiguous share of `batches`; returns the global mean of every metric over batches
    (identical on all ranks)."""
    env = env or DistEnv(0, 0, 1, torch.device("cuda:0"), None)
    lo, hi = shard_range(len(batches), env.rank, env.world)
    sums = [0.0] * len(metrics)
    with torch.no_grad():
        for batch in batches[lo:hi]:
            batch = move_batch_to_device(batch, env.device, [k for k in device_tensors if k in batch])
            batch.update(model(**batch))                        # inferencer.py:117-118
            for i, met in enumerate(metrics):                   # inferencer.py:125-126
                sums[i] += float(met(**batch))
    tot = env.sum_over_ranks(sums + [float(hi - lo)])
    return {met.name: tot[i] / max(tot[-1], 1.0) for i, met in enumerate(metrics)}
