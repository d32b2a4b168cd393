"""One training step around the HIP model (BASELINE config 4): the per-batch semantics of the reference's
Trainer.process_batch (src/trainer/trainer.py:13-59) with one-process-per-GPU data parallelism added.

  zero_grad -> outputs = model(**batch) -> loss = criterion(**batch) -> loss.backward() -> [all-reduce of gradients]
  -> clip_grad_norm_(max_grad_norm) -> optimizer.step() -> lr_scheduler.step()          (trainer.py:38-51)

Every piece runs on the device without host synchronisation: the model's forward/backward are libdptnav kernels
(model.py), the loss is `metrics.SiSNRWavLoss` (dptnav_pit_sisnr_loss: permutation resolved on the device, gradient
written by the forward), the clip and AdamW are `optim.clip_grad_norm_` / `optim.FusedAdamW` over the model's ONE flat
gradient tensor (17.8 MB), which is also what the data-parallel all-reduce reduces in place: on xGMI (7 point-to-point
links per GPU) one large collective beats 228 small ones; the global-norm clip runs after the all-reduce, as the
reference's clip does on the full gradient (base_trainer.py:383-391).  The returned loss / gradient norm are 0-dim
DEVICE tensors: reading them (logging, trainer.py:55) is the caller's only synchronisation, not one per step.
PIT is batch level: every rank resolves the permutation on its own 16 mixtures (SURVEY.md 8e caveat).
Stock torch.optim optimizers and torch.nn.utils.clip_grad_norm_ keep working (the parameters and .grad are ordinary
tensors); `train_step` picks the fused clip whenever every `.grad` is a view of the drop-in's flat gradient and torch's otherwise
(frozen parameters, gradient accumulation).
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch
from torch import nn

from .metrics import SiSNRWavLoss  # noqa: F401  (re-export: the criterion of the training step)
from .optim import FusedAdamW, clip_grad_norm_, flat_grad_or_none  # noqa: F401
from .parallel import DistEnv


def allreduce_gradients(model: nn.Module, env: Optional[DistEnv]) -> str:
    """Average the gradients over ranks with ONE collective on a flat bucket (no-op for a single process).  Returns which
    way it went: "single", "flat-in-place" (the drop-in's own flat gradient tensor, no copies) or "flat-copy"."""
    if env is None or not env.active:
        return "single"
    import torch.distributed as dist
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    # the drop-in hands autograd views of one flat tensor (model._SeparateFn.backward): reduce that tensor in place
    flat = getattr(model, "_flat_grad", None)
    if flat is not None and grads and all(g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(env.world)
        return "flat-in-place"
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(env.world)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
    return "flat-copy"


def train_step(model: nn.Module, batch: Dict[str, object], criterion, optimizer, max_grad_norm: Optional[float] = 10.0,
               lr_scheduler=None, env: Optional[DistEnv] = None) -> Mapping[str, torch.Tensor]:
    """-> {"loss", "grad_norm"} as 0-dim device tensors (no .item() here: trainer.py:55 reads them when it logs)."""
    optimizer.zero_grad()
    outputs = model(**batch)
    batch.update(outputs)
    loss = criterion(**batch)["loss"]
    loss.backward()
    allreduce_gradients(model, env)
    norm = None
    if max_grad_norm is not None:
        # fused clip (two launches on the flat tensor) only when every .grad is a view of it; frozen parameters
        # (train.py:51 passes filter(requires_grad)), accumulated gradients or foreign modules take torch's clip
        if getattr(model, "_flat_grad", None) is not None and flat_grad_or_none(model) is not None:
            norm = clip_grad_norm_(model, max_grad_norm)
        else:
            norm = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_grad_norm)
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return {"loss": loss.detach(), "grad_norm": norm if norm is not None else torch.full((), float("nan"))}
