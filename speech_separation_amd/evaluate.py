"""Data-parallel evaluation loop: the per-batch semantics of the reference's Inferencer.process_batch
(src/trainer/inferencer.py:98-127) and MetricTracker (src/metrics/tracker.py:29-42: running mean of per-batch
values), sharded over ranks by speech_separation_amd.parallel with one SUM all-reduce at the end.
`evaluate` takes ready-made batches; `run_inference` is the whole input-to-files pipeline."""
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


def run_inference(model: Callable[..., Mapping[str, torch.Tensor]], entries: List[Mapping[str, Optional[str]]], batch_size: int,
                  metrics: List[object], save_dir: Optional[str] = None, device="cuda:0", workers: int = 8,
                  prefetch: int = 3, target_sr: Optional[int] = None, env: Optional[DistEnv] = None):
    """The reference's Inferencer._inference_part (src/trainer/inferencer.py:169-202) over a dataset index, as a pipeline
    that keeps up with a forward of hundreds of mixtures per second:

        loader threads (WAV + .npz decode, io.load_item)  ->  PinnedBatcher (collate into pinned memory, async H2D)
        ->  model(**batch)  ->  metrics  ->  PredictionWriter (async D2H, one <stem>.pth per item on a worker thread)

    `entries` are index dicts with the reference's keys (mix_wav_path, s1_wav_path, s2_wav_path, s1_embedding_path,
    s2_embedding_path; base_dataset.py:70-98).  Ranks take contiguous runs of whole batches (no data-path collective).
    Returns (logs, stats): logs = mean over batches of every metric (MetricTracker semantics, global over ranks);
    stats = {"items", "seconds", "items_per_s"} of THIS rank's loop, wall clock from the first load to the last file."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    from .io import PinnedBatcher, PredictionWriter, load_item, save_predictions
    env = env or DistEnv(0, 0, 1, torch.device(device), None)
    dev = env.device
    nb = (len(entries) + batch_size - 1) // batch_size
    lo, hi = shard_range(nb, env.rank, env.world)
    mine = [entries[b * batch_size:(b + 1) * batch_size] for b in range(lo, hi)]
    batcher = PinnedBatcher(dev)
    cuda = dev.type == "cuda"          # (a CPU device only exists for the host-logic rehearsals in tests/: gloo ranks, a stand-in model)
    writer = PredictionWriter(dev) if save_dir is not None and cuda else None
    cpu_paths: List[str] = []
    sums = [0.0] * len(metrics)
    met_means = []        # per batch: the metrics' device-side means (read once, after the loop: no stall per batch)
    n_items = 0
    # The launching thread shares the interpreter lock with the loader threads and the writer: at CPython's default switch interval
    # (5 ms) every one of its hand-overs can cost that long while a loader holds the lock in pure-Python code, and a batch has
    # ~28 ms in all.  A shorter interval for the duration of the loop (restored below).
    import contextlib
    import sys

    @contextlib.contextmanager
    def short_switch_interval(seconds=2e-4):
        old = sys.getswitchinterval()
        sys.setswitchinterval(min(old, seconds))
        try:
            yield
        finally:
            sys.setswitchinterval(old)

    t0 = time.perf_counter()
    with short_switch_interval(), ThreadPoolExecutor(max_workers=max(1, workers)) as pool, torch.no_grad():
        load = lambda e: load_item(e, target_sr)
        pending = [[pool.submit(load, e) for e in b] for b in mine[:prefetch]]
        for i in range(len(mine)):
            if i + prefetch < len(mine):
                pending.append([pool.submit(load, e) for e in mine[i + prefetch]])
            items = [f.result() for f in pending.pop(0)]
            batch = batcher.to_device(items)
            batch.update(model(**batch))                        # inferencer.py:117-118
            if batch.get("s1") is not None:                     # inferencer.py:125-126
                if all(hasattr(met, "enqueue") for met in metrics):
                    met_means.append([met.enqueue(**batch) for met in metrics])
                else:
                    for j, met in enumerate(metrics):
                        sums[j] += float(met(**batch))
            if writer is not None:
                writer.submit(batch, save_dir)                  # inferencer.py:128-147
            elif save_dir is not None:
                cpu_paths += save_predictions(batch, save_dir)
            n_items += len(items)
    for per_batch in met_means:
        for j, (met, means) in enumerate(zip(metrics, per_batch)):
            sums[j] += float(met.resolve(means.cpu()))
    paths = writer.close() if writer is not None else cpu_paths
    if cuda:
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tot = env.sum_over_ranks(sums + [float(hi - lo)])
    logs = {met.name: tot[j] / max(tot[-1], 1.0) for j, met in enumerate(metrics)}
    return logs, {"items": n_items, "seconds": dt, "items_per_s": n_items / max(dt, 1e-9), "files": len(paths)}
