"""PIT SI-SNR loss (forward + backward) and SI-SNR(i) metrics computed on the GPU.

Mirrors (does not import) the reference interfaces that consume the model's outputs:
  * ``SiSNRWavLoss()(**batch) -> {"loss": tensor}``          src/loss/ss_losses.py:117-130 (+ BaseSSLoss :21-26)
  * ``SISNRiMetric(name=..., device=...)(**batch) -> value``  src/metrics/si_snri.py:7-30
  * ``SISNRMetric(name=..., device=...)(**batch) -> value``   src/metrics/si_snr.py:6-12
Both resolve the speaker permutation at BATCH level (compare the two batch means), exactly as the reference does
(ss_losses.py:21-25, base_metric.py:57-60) -- this is not per-utterance PIT.

The reference needs >= 6 ``.item()`` syncs per batch for the metrics (base_metric.py:53-56, si_snri.py:25-26); here the
per-item statistics come from ``dptnav_sisnr_pairs`` (one launch) and the 12*B numbers are reduced on the host with ONE
device->host copy.  The LOSS never touches the host: ``dptnav_pit_sisnr_loss`` resolves the permutation on the device,
returns the loss as a 0-dim device tensor and leaves d loss / d prediction ready for ``loss.backward()`` (two launches
instead of ~40 PyTorch kernels and a tensor->bool conversion).
"""
from __future__ import annotations

from typing import Dict

import torch

from .engine import DptnEngine
from .spec import DPTN_AV

_ENGINES: Dict[torch.device, DptnEngine] = {}


def _engine(device: torch.device) -> DptnEngine:
    if device.type != "cuda":
        raise RuntimeError("speech_separation_amd.metrics computes on an AMD GPU through libdptnav; there is no CPU path")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _ENGINES:
        _ENGINES[device] = DptnEngine(DPTN_AV, device)   # the statistics kernel needs a handle, not weights
    return _ENGINES[device]


def pair_statistics(s1_pred, s2_pred, s1, s2, mix) -> torch.Tensor:
    """(6,2) CPU tensor of BATCH MEANS: rows = (p1,s1) (p1,s2) (p2,s1) (p2,s2) (mix,s1) (mix,s2);
    columns = (SI-SNR dB, reference loss term).  One launch, one sync."""
    stats = _engine(mix.device).sisnr_pairs(s1_pred, s2_pred, s1, s2, mix)
    return stats.double().mean(0).cpu()


class _PitSisnrFn(torch.autograd.Function):
    """loss = BaseSSLoss(SiSNRLoss) (ss_losses.py:21-26,100-114); the forward launch pair already writes the gradient."""

    @staticmethod
    def forward(ctx, s1_pred, s2_pred, s1, s2):
        d1, d2, out = _engine(s1_pred.device).pit_sisnr_loss(s1_pred, s2_pred, s1, s2)
        ctx.save_for_backward(d1, d2)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _g_stats):
        d1, d2 = ctx.saved_tensors
        return g * d1, g * d2, None, None


class SiSNRWavLoss(torch.nn.Module):
    """The reference's PIT SI-SNR loss: ``SiSNRWavLoss()(**batch) -> {"loss": 0-dim tensor}`` (ss_losses.py:117-130),
    differentiable w.r.t. s1_pred / s2_pred, no host synchronisation.  ``self.last`` keeps the device tensor
    [loss, permutation, loss perm 0, loss perm 1] of the latest call for logging."""

    def forward(self, s1_pred, s2_pred, s1, s2, **batch):
        loss, self.last = _PitSisnrFn.apply(s1_pred, s2_pred, s1, s2)
        return {"loss": loss}


class SISNRMetric:
    """``metric(**batch) -> value`` as the reference's (one launch, one device->host copy).  For a loop that must not stall
    on every batch the call is also available in two halves: ``enqueue(**batch)`` launches the statistics kernel and
    returns the (6,2) batch means as a DEVICE tensor without synchronising, ``resolve(means.cpu())`` finishes on the
    host (evaluate.run_inference reads all batches' means once, at the end)."""

    def __init__(self, name=None, device="cuda", lower_better=False, *args, **kwargs):
        self.name = name if name is not None else type(self).__name__
        self.pick = min if lower_better else max

    def _pit(self, m):
        return self.pick(float((m[0] + m[3]) / 2), float((m[1] + m[2]) / 2))

    def enqueue(self, s1_pred, s2_pred, s1, s2, mix=None, **batch) -> torch.Tensor:
        stats = _engine(s1_pred.device).sisnr_pairs(s1_pred, s2_pred, s1, s2, s1_pred if mix is None else mix)
        return stats.double().mean(0)

    def resolve(self, means: torch.Tensor):
        return self._pit(means[:, 0])

    def __call__(self, s1_pred, s2_pred, s1, s2, mix=None, **batch):
        return self.resolve(self.enqueue(s1_pred, s2_pred, s1, s2, mix).cpu())


class SISNRiMetric(SISNRMetric):
    def resolve(self, means: torch.Tensor):
        m = means[:, 0]
        return torch.tensor(self._pit(m) - float((m[4] + m[5]) / 2))   # 0-dim tensor like the reference (float - tensor)

    def __call__(self, s1_pred, s2_pred, s1, s2, mix, **batch):
        return self.resolve(self.enqueue(s1_pred, s2_pred, s1, s2, mix).cpu())
