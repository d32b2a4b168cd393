"""Callers and data formats either side of the hot path: embedding / tensor loading, batch
collation with pinned staging + asynchronous H2D, and the per-item prediction writer.

Mirrors (does not import) the reference's formats and semantics:
  * ``load_object``        src/datasets/base_dataset.py:188-205   .npy / .npz (first array; make_embeddings.py:69 writes
                                                                   ``np.savez_compressed(embedding=(512,50) f32)``) / .pt|.pth,
                                                                   returned with a leading batch dim of 1
  * ``collate``            src/datasets/collate.py:4-46           tensors concatenated on dim 0, ``audio_path`` as a list,
                                                                   a key whose first item is None stays None
  * ``save_predictions``   src/trainer/inferencer.py:128-147      one ``<stem>.pth`` per item holding 1-D tensors
                                                                   {s1_pred, s2_pred[, s1_true, s2_true]}
Once the forward runs at hundreds of mixtures per second these become the bottleneck of an evaluation, so the batch is
assembled directly in reusable PINNED host buffers (one non-blocking copy per tensor, on a side stream, overlapped with
the previous batch's forward) and the predictions come back in ONE device->host copy per tensor instead of 4·B clones.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

TENSOR_KEYS = ["mix_spectrogram", "complex_spectrogram", "s1_spectrogram", "s2_spectrogram", "s1_video", "s2_video",
               "s1_embedding", "s2_embedding", "mix", "s1", "s2"]
LIST_KEYS = ["audio_path"]


def load_object(path: str) -> torch.Tensor:
    if path.endswith(".npy"):
        obj = torch.from_numpy(np.load(path))
    elif path.endswith(".npz"):
        with np.load(path) as data:
            obj = torch.from_numpy(data[next(iter(data))])
    elif path.endswith((".pt", ".pth")):
        obj = torch.load(path)
    else:
        raise ValueError(f"unsupported object file: {path}")
    return obj.unsqueeze(0)


def collate(items: Sequence[Mapping[str, object]]) -> Dict[str, object]:
    batch: Dict[str, object] = {}
    for key in TENSOR_KEYS + LIST_KEYS:
        if key not in items[0]:
            continue
        if items[0][key] is None:
            batch[key] = None
            continue
        vals = [it[key] for it in items]
        batch[key] = torch.cat(vals, dim=0) if key in TENSOR_KEYS else vals
    return batch


class PinnedBatcher:
    """collate() into reusable pinned buffers + non-blocking H2D on a side stream.

    ``to_device(items)`` returns the batch dict with device tensors; the copies are ordered before later work on the
    CURRENT stream by an event, so the caller can use the tensors as usual."""

    def __init__(self, device, device_tensors: Sequence[str] = ("mix", "s1", "s2", "s1_embedding", "s2_embedding")):
        self.device = torch.device(device)
        self.device_tensors = list(device_tensors)
        self._pinned: Dict[str, torch.Tensor] = {}
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _stage(self, key: str, vals: List[torch.Tensor]) -> torch.Tensor:
        shape = (sum(v.shape[0] for v in vals),) + tuple(vals[0].shape[1:])
        buf = self._pinned.get(key)
        if buf is None or tuple(buf.shape) != shape or buf.dtype != vals[0].dtype:
            buf = torch.empty(shape, dtype=vals[0].dtype, pin_memory=self.device.type == "cuda")
            self._pinned[key] = buf
        o = 0
        for v in vals:                       # the concatenation itself: written straight into pinned memory
            buf[o:o + v.shape[0]].copy_(v)
            o += v.shape[0]
        return buf

    def to_device(self, items: Sequence[Mapping[str, object]]) -> Dict[str, object]:
        batch: Dict[str, object] = {}
        staged = {}
        for key in TENSOR_KEYS + LIST_KEYS:
            if key not in items[0]:
                continue
            if items[0][key] is None:
                batch[key] = None
            elif key in LIST_KEYS:
                batch[key] = [it[key] for it in items]
            elif key in self.device_tensors:
                staged[key] = self._stage(key, [it[key] for it in items])
            else:
                batch[key] = torch.cat([it[key] for it in items], dim=0)
        if self._stream is None:
            batch.update({k: v.clone() for k, v in staged.items()})
            return batch
        with torch.cuda.stream(self._stream):
            for k, v in staged.items():
                batch[k] = v.to(self.device, non_blocking=True)
        torch.cuda.current_stream(self.device).wait_stream(self._stream)
        return batch


def save_predictions(batch: Mapping[str, object], out_dir: str) -> List[str]:
    """Write ``<stem>.pth`` per item (inferencer.py:128-147 format).  One D2H copy per tensor, not per item."""
    os.makedirs(out_dir, exist_ok=True)
    host = {k: batch[k].detach().to("cpu") for k in ("s1_pred", "s2_pred")}
    if batch.get("s1") is not None:
        host["s1_true"] = batch["s1"].detach().to("cpu")
        host["s2_true"] = batch["s2"].detach().to("cpu")
    paths = []
    for i, ap in enumerate(batch["audio_path"]):
        p = os.path.join(out_dir, f"{Path(ap).stem}.pth")
        torch.save({k: v[i].clone() for k, v in host.items()}, p)
        paths.append(p)
    return paths
