"""One training step around the HIP model (BASELINE config 4): the per-batch semantics of the reference's
Trainer.process_batch (src/trainer/trainer.py:13-59) with one-process-per-GPU data parallelism added.

  zero_grad -> outputs = model(**batch) -> loss = criterion(**batch) -> loss.backward() -> [all-reduce of gradients]
  -> clip_grad_norm_(max_grad_norm) -> optimizer.step() -> lr_scheduler.step()          (trainer.py:38-51)

The model's forward/backward are libdptnav kernels (model.py); the loss below is the reference's SI-SNR PIT loss
restated with the same PyTorch operators (src/loss/ss_losses.py:21-26,100-114) -- in the reference this stays
`src.loss.SiSNRWavLoss`, unchanged.  Gradients of all 228 tensors are flattened into ONE bucket (17.8 MB) and
all-reduced over RCCL: on xGMI (7 point-to-point links per GPU) one large collective beats 228 small ones; the
global-norm clip runs after the all-reduce, as the reference's clip does on the full gradient (base_trainer.py:383-391).
PIT is batch level: every rank resolves the permutation on its own 16 mixtures (SURVEY.md 8e caveat).
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch
from torch import nn

from .parallel import DistEnv


class SiSNRWavLoss(nn.Module):
    """-20 log10(|a t|^2 / |p - a t|^2) on zero-mean signals, batch mean, batch-level PIT (ss_losses.py:21-26,100-114)."""

    @staticmethod
    def _one(pred, gt):
        pred = pred - pred.mean(-1, keepdim=True)
        gt = gt - gt.mean(-1, keepdim=True)
        scale = (gt * pred).sum(-1, keepdim=True) / (gt * gt).sum(-1, keepdim=True)
        st = scale * gt
        return (-20 * torch.log10((st * st).sum(-1) / ((pred - st) ** 2).sum(-1))).mean()

    def forward(self, s1_pred, s2_pred, s1, s2, **batch):
        p1 = (self._one(s1_pred, s1) + self._one(s2_pred, s2)) / 2
        p2 = (self._one(s1_pred, s2) + self._one(s2_pred, s1)) / 2
        return {"loss": p2 if p2 < p1 else p1}


def allreduce_gradients(model: nn.Module, env: Optional[DistEnv]) -> None:
    """Average the gradients over ranks with ONE collective on a flat bucket (no-op for a single process)."""
    if env is None or env.world == 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    # the drop-in hands autograd views of one flat tensor (model._SeparateFn.backward): reduce that tensor in place
    flat = getattr(model, "_flat_grad", None)
    if flat is not None and grads and all(g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(env.world)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(env.world)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def train_step(model: nn.Module, batch: Dict[str, object], criterion, optimizer, max_grad_norm: Optional[float] = 10.0,
               lr_scheduler=None, env: Optional[DistEnv] = None) -> Mapping[str, float]:
    optimizer.zero_grad()
    outputs = model(**batch)
    batch.update(outputs)
    loss = criterion(**batch)["loss"]
    loss.backward()
    allreduce_gradients(model, env)
    norm = None
    if max_grad_norm is not None:
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return {"loss": float(loss.detach()), "grad_norm": float(norm) if norm is not None else float("nan")}
