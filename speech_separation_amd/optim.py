"""Gradient clipping and AdamW for the libdptnav drop-in models, on the device with no host synchronisation.

Mirrors (does not import) what the reference's training step calls around the model:
  * ``clip_grad_norm_(self.model.parameters(), max_grad_norm)``      src/trainer/base_trainer.py:383-391
  * ``torch.optim.AdamW(params, lr=1e-3)`` (Hydra ``optimizer`` node)  src/configs/dptn_wav_av.yaml:9-11, train.py:51-52
    -- select with ``optimizer._target_=speech_separation_amd.FusedAdamW``

The drop-in model hands autograd ONE flat gradient tensor (``model._flat_grad``, layout ``dptnav_flat_offset``); the clip
is two launches over it (fixed-order sum of squares, scale) and the optimizer step four (64 tensors each), instead of the
~10 multi-tensor launches + several hundred tiny ones of the stock path.  ``state_dict()`` has torch.optim.AdamW's
format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so base_trainer.py:476-478,528-535 checkpoints load
either way.  No CPU path: a model that has not run a backward on the GPU raises.
"""
from __future__ import annotations

from typing import Iterable, Optional, Union

import torch
from torch import nn


def _owner(params) -> "nn.Module":
    params = list(params)
    if not params:
        raise ValueError("no parameters")
    ref = getattr(params[0], "_dptnav_owner", None)
    model = ref() if ref is not None else None
    if model is None:
        raise RuntimeError("these parameters do not belong to a speech_separation_amd model (no libdptnav engine to run "
                           "the fused step on); use torch.optim.AdamW / torch.nn.utils.clip_grad_norm_ for other modules")
    mine = list(model.parameters())
    if len(mine) != len(params) or any(a is not b for a, b in zip(mine, params)):
        raise RuntimeError("the fused clip / optimizer step covers ALL parameters of one model in state_dict order "
                           "(the reference trains every parameter, train.py:51)")
    return model


def _flat_grad_problem(model: "nn.Module") -> Optional[str]:
    """None when every parameter's .grad is the matching view of `model._flat_grad` (the fused clip / step may run on the
    flat tensor); otherwise the reason why not, naming the first offending parameter."""
    flat = getattr(model, "_flat_grad", None)
    if flat is None:
        return "no gradient yet: run loss.backward() through the model first"
    eng = model._engine
    base, offs = flat.data_ptr(), eng._grad_offsets
    for key, p in model.named_parameters():
        if not p.requires_grad:
            return (f"{key} is frozen (requires_grad=False): the fused clip / optimizer step covers ALL parameters of the "
                    f"model; use torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW on the trainable subset")
        if p.grad is None:
            return (f"{key}.grad is None (zero_grad(set_to_none=True) after the backward, or the parameter did not take part "
                    f"in it): nothing to clip / step")
        if p.grad.data_ptr() != base + 4 * offs[key]:
            return (f"{key}.grad is not a view of the model's flat gradient (it was accumulated over several backward passes, "
                    f"or kept by zero_grad(set_to_none=False)): the fused clip / step needs zero_grad() with its default "
                    f"set_to_none=True before every backward")
    return None


def _flat_grad(model: "nn.Module") -> torch.Tensor:
    why = _flat_grad_problem(model)
    if why is not None:
        raise RuntimeError(why)
    return model._flat_grad


def flat_grad_or_none(model: "nn.Module") -> Optional[torch.Tensor]:
    """The flat gradient tensor if the fused clip / step can run on it, else None (train_step then takes the stock path)."""
    return None if _flat_grad_problem(model) is not None else model._flat_grad


def clip_grad_norm_(parameters: Union["nn.Module", Iterable[torch.Tensor]], max_norm: Optional[float]) -> torch.Tensor:
    """Global L2-norm clip of a drop-in model's gradients, in place; returns the norm before clipping as a 0-dim DEVICE
    tensor (like torch's, which base_trainer.py:388 discards) -- reading it is the caller's synchronisation, not ours."""
    model = parameters if isinstance(parameters, nn.Module) else _owner(parameters)
    flat = _flat_grad(model)
    return model._get_engine(flat.device).grad_clip(flat, max_norm)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction; amsgrad / maximize off) in one pass over the
    flat gradient + state buffers.  The learning rate is read from the param group every step, so OneCycleLR
    (dptn_wav_av.yaml:12-18) drives it unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False,
                 **unused):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdamW: amsgrad / maximize are not built (the reference uses neither)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW: one parameter group (train.py:51 passes one)")
        self._model_ref = None
        self._m = self._v = None
        self._step = 0

    def _model(self):
        if self._model_ref is None:
            import weakref
            self._model_ref = weakref.ref(_owner(self.param_groups[0]["params"]))
        model = self._model_ref()
        if model is None:
            raise RuntimeError("the model this optimizer was built for is gone")
        return model

    def _ensure_state(self, eng):
        if self._m is not None and self._m.device == eng.device:
            return
        old = (self._m, self._v)
        self._m = torch.zeros(eng.flat_numel(), device=eng.device)
        self._v = torch.zeros(eng.flat_numel(), device=eng.device)
        if old[0] is not None:                       # the model moved to another device
            self._m.copy_(old[0])
            self._v.copy_(old[1])
        offs = eng.flat_offsets()
        for key, p in self._model().named_parameters():
            o, n = offs[key], p.numel()
            self.state[p] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[o:o + n].view_as(p),
                             "exp_avg_sq": self._v[o:o + n].view_as(p)}

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedAdamW: closures are not supported")
        model = self._model()
        flat = _flat_grad(model)
        eng = model._get_engine(flat.device)        # re-binds the parameter pointers if they moved
        self._ensure_state(eng)
        g = self.param_groups[0]
        self._step += 1
        eng.adamw_step(flat, self._m, self._v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step)

    # ---- checkpoint compatibility with torch.optim.AdamW (base_trainer.py:476-478, 528-535) ------------------------------
    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # per-parameter tensors, cast to the parameters' device
        loaded = {p: dict(st) for p, st in self.state.items()}
        if not loaded:
            return
        model = self._model()
        p0 = next(iter(loaded))
        eng = model._get_engine(p0.device)
        self._m = self._v = None
        self._step = int(float(loaded[p0]["step"]))
        self.state.clear()
        self._ensure_state(eng)
        for p, st in loaded.items():
            self.state[p]["exp_avg"].copy_(st["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
