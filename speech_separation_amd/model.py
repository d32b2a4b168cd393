"""Drop-in ``nn.Module``s for the reference's DPTN raw-waveform separators, backed by libdptnav.

Mirrors (does not import) the reference interface for this path:
  * ``DPTNAVWavEncDec(num_features, video_emb_size, hidden_video, kernel_size_enc, hidden_dim, num_blocks,
    chunk_size, step_size, num_heads, dropout, bidir)``            src/model/dptn_wav.py:137-150
  * ``DPTNWavEncDec(num_features, kernel_size_enc, ...)``           src/model/dptn_wav.py:72-83
  * ``forward(mix, s1_embedding, s2_embedding, **batch) -> {"s1_pred","s2_pred"}``  dptn_wav.py:171,194
    -- called as ``self.model(**batch)`` by src/trainer/trainer.py:40 and inferencer.py:117, so unknown
    batch keys (mix_spectrogram, s1, s2, paths, ...) must be accepted and ignored.
  * ``state_dict()`` keys/shapes == the reference's, so ``model_best.pth`` loads
    through base_trainer.py:539-560 unchanged.
  * ``str(model)`` ends with the two parameter-count lines of dptn_wav.py:196-207.

Select it from the reference's Hydra CLI with ``model._target_=speech_separation_amd.DPTNAVWavEncDec``
(INTEGRATION.md).  The forward AND the backward of the model run ONLY on the HIP library (a torch.autograd.Function
hands autograd the parameter gradients computed by dptnav_train_backward): no PyTorch operators are used for the
model's compute and there is no CPU fallback -- a missing extension or a CPU tensor raises.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, Optional

import torch
from torch import nn

from .engine import DptnEngine
from .spec import DPTNConfig, state_dict_spec


class _Node(nn.Module):
    """Pure parameter container (never called): gives nested state_dict keys such as
    ``dprnn.model.0.intra_chunk_block.mha.out_proj.weight``."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute happens in libdptnav, not in torch")


def _init_like_torch(key: str, p: torch.Tensor, cfg: DPTNConfig) -> None:
    """Default initialisers of the stock modules the reference instantiates (so a from-scratch run
    starts from the same distributions): nn.LSTM U(+-1/sqrt(H)); nn.Linear/Conv kaiming_uniform(a=sqrt 5)
    == U(+-1/sqrt(fan_in)) with the matching bias bound; nn.MultiheadAttention xavier_uniform in_proj,
    zero biases; LayerNorm 1/0; PReLU 0.25; gate ~ N(0,1) (dptn_wav.py:155)."""
    leaf2 = ".".join(key.split(".")[-2:])
    with torch.no_grad():
        if key == "gate":
            p.normal_()
        elif key.endswith("speakers_separation.0.weight"):
            p.fill_(0.25)
        elif ".rnn." in key:
            b = 1.0 / math.sqrt(cfg.hidden_dim)
            p.uniform_(-b, b)
        elif leaf2 in ("ln1.weight", "ln2.weight", "video_ln.weight", "norm1d.weight"):
            p.fill_(1.0)
        elif leaf2 in ("ln1.bias", "ln2.bias", "video_ln.bias", "norm1d.bias", "mha.in_proj_bias", "out_proj.bias"):
            p.zero_()
        elif leaf2 == "mha.in_proj_weight":
            nn.init.xavier_uniform_(p)
        elif key.endswith("weight"):
            fan_in = p[0].numel()
            b = 1.0 / math.sqrt(fan_in)
            p.uniform_(-b, b)
        else:  # biases of Linear / Conv: U(+-1/sqrt(fan_in of the matching weight))
            fan_in = {"visual_compression.bias": cfg.video_emb_size, "ffn.1.bias": None, "fc.bias": None}.get(
                leaf2, cfg.num_features)
            if fan_in is None:
                fan_in = cfg.hidden_dim * (2 if (cfg.bidir or "intra_chunk_block" in key) else 1)
            b = 1.0 / math.sqrt(fan_in)
            p.uniform_(-b, b)


class _SeparateFn(torch.autograd.Function):
    """Model part of the training step: forward records a tape in libdptnav, backward turns d loss / d predictions into
    the gradient of every parameter (dptnav_train_backward).  Loss, clipping and optimizer stay ordinary PyTorch code
    of the caller (the reference's trainer.py:43-51)."""

    @staticmethod
    def forward(ctx, module, mix, e1, e2, *params):
        eng = module._get_engine(mix.device)
        if getattr(eng, "_grads", None) is None:
            eng.bind_grads()
        ppm, seed = module._arm_dropout(eng)
        s1, s2, tape = eng.train_forward(mix, e1, e2)
        ctx.module, ctx.tape, ctx.inputs, ctx.drop = module, tape, (mix, e1, e2), (ppm, seed)
        return s1, s2

    @staticmethod
    def backward(ctx, d_s1, d_s2):
        eng = ctx.module._engine
        mix, e1, e2 = ctx.inputs
        zeros = lambda g: torch.zeros_like(mix) if g is None else g.contiguous()
        eng.set_option("dropout_ppm", ctx.drop[0])
        eng.set_option("dropout_seed", ctx.drop[1])
        eng.train_backward(mix, e1, e2, zeros(d_s1), zeros(d_s2), ctx.tape)
        ctx.tape = None
        # the library's gradient buffers are reused by the next step: hand autograd its own copy -- ONE flat copy whose
        # per-parameter views autograd adopts as .grad without copying (train.allreduce_gradients finds the flat
        # tensor again through `_flat_grad`)
        flat = eng._grads_flat.clone()
        ctx.module._flat_grad = flat
        outs = []
        for k, shape in eng.slots:
            o = eng._grad_offsets[k]
            outs.append(flat[o:o + eng._grads[k].numel()].view(*shape))
        return (None, None, None, None) + tuple(outs)


class _DPTNBase(nn.Module):
    def __init__(self, cfg: DPTNConfig):
        super().__init__()
        self.cfg = cfg
        for key, shape in state_dict_spec(cfg):
            parts = key.split(".")
            node: nn.Module = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            p = nn.Parameter(torch.empty(*shape))
            node.register_parameter(parts[-1], p)
            _init_like_torch(key, p, cfg)
            p._dptnav_owner = weakref.ref(self)      # lets optim.FusedAdamW / clip_grad_norm_ find the engine
        self._engine: Optional[DptnEngine] = None

    # -- engine management -------------------------------------------------------------------
    def _params(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())

    def _get_engine(self, device: torch.device) -> DptnEngine:
        if device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} computes only on an AMD GPU through libdptnav "
                               f"(got a {device} tensor); there is no CPU/PyTorch fallback")
        eng = self._engine
        if eng is None or eng.device != device:
            eng = DptnEngine(self.cfg, device)
            self._engine = eng
        params = self._params()
        for k, p in params.items():
            if p.device != device:
                raise RuntimeError(f"parameter {k} is on {p.device} but the input is on {device}: call model.to(device)")
        if not eng.bound_to(params):
            eng.bind(params)  # borrows the nn.Parameter storages (optimizer / load_state_dict stay in charge)
        return eng

    def _arm_dropout(self, eng: DptnEngine):
        """Train-mode attention dropout (dptn.py:16-21): a fresh seed per call, kept for this call's backward.  Data-parallel
        ranks share torch's seed, so the rank is mixed in to keep their masks apart."""
        ppm = int(round(self.cfg.dropout * 1e6)) if self.training else 0
        self._drop_step = getattr(self, "_drop_step", 0) + 1
        rank = int(os.environ.get("RANK", "0"))
        seed = (torch.initial_seed() * 2654435761 + self._drop_step * 40503 + rank * 0x632BE5AB) & 0x7FFFFFFF
        eng.set_option("dropout_ppm", ppm)
        eng.set_option("dropout_seed", seed)
        return ppm, seed

    def _train_kernels_built(self) -> bool:
        return self.cfg.num_features in (128, 64)        # DPTN and DPRNN blocks, bidirectional or not

    def _run(self, mix, e1, e2):
        cont = lambda t: None if t is None else t.contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if not self._train_kernels_built():
                raise NotImplementedError("the training step (backward kernels) is built for num_features in {128, 64}; use "
                                          "torch.no_grad() here")
            s1, s2 = _SeparateFn.apply(self, mix.contiguous(), cont(e1), cont(e2), *self.parameters())
            return {"s1_pred": s1, "s2_pred": s2}
        eng = self._get_engine(mix.device)
        # model.train() under torch.no_grad(): the reference still applies attention dropout (nn.MultiheadAttention looks at
        # self.training only).  The dropout lives in the training kernels, so take that path and drop the tape; where it
        # is not built, say so instead of silently returning the eval-mode result.  DPRNN blocks have no dropout.
        if self.training and self.cfg.arch == "dptn" and self.cfg.dropout > 0:
            if not self._train_kernels_built():
                raise NotImplementedError("train-mode forward (attention dropout) is built for num_features in {128, 64} "
                                          "only: call model.eval() for inference")
            self._arm_dropout(eng)
            s1, s2, _tape = eng.train_forward(mix.contiguous(), cont(e1), cont(e2))
            return {"s1_pred": s1, "s2_pred": s2}
        s1, s2 = eng.forward(mix, e1, e2)
        return {"s1_pred": s1, "s2_pred": s2}

    def __str__(self):
        all_parameters = sum(p.numel() for p in self.parameters())
        trainable_parameters = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return (super().__str__() + f"\nAll parameters: {all_parameters}"
                + f"\nTrainable parameters: {trainable_parameters}")


class DPTNAVWavEncDec(_DPTNBase):
    """Audio-visual DPTN (BASELINE configs 3/4) -- same constructor as the reference class of that name."""

    def __init__(self, num_features=64, video_emb_size=1024, hidden_video=128, kernel_size_enc=2, hidden_dim=32,
                 num_blocks=6, chunk_size=10, step_size=5, num_heads=4, dropout=0.1, bidir=True):
        super().__init__(DPTNConfig(num_features=num_features, video_emb_size=video_emb_size,
                                    hidden_video=hidden_video, kernel_size_enc=kernel_size_enc,
                                    hidden_dim=hidden_dim, num_blocks=num_blocks, chunk_size=chunk_size,
                                    step_size=step_size, num_heads=num_heads, dropout=dropout, bidir=bool(bidir),
                                    audio_only=False))

    def forward(self, mix, s1_embedding, s2_embedding, **batch):
        return self._run(mix, s1_embedding, s2_embedding)


class DPTNWavEncDec(_DPTNBase):
    """Audio-only DPTN (BASELINE config 2) -- same constructor as the reference class of that name."""

    def __init__(self, num_features=64, kernel_size_enc=2, hidden_dim=32, num_blocks=6, chunk_size=10, step_size=5,
                 num_heads=4, dropout=0.1, bidir=True):
        super().__init__(DPTNConfig(num_features=num_features, kernel_size_enc=kernel_size_enc,
                                    hidden_dim=hidden_dim, num_blocks=num_blocks, chunk_size=chunk_size,
                                    step_size=step_size, num_heads=num_heads, dropout=dropout, bidir=bool(bidir),
                                    audio_only=True))

    def forward(self, mix, **batch):
        return self._run(mix, None, None)


class DPRNNEncDec(_DPTNBase):
    """Audio-only DPRNN (backbone of BASELINE config 5) -- same constructor as the reference class of that name
    (src/model/dprnn.py:238-247); forward(mix, **batch) as dprnn.py:260."""

    def __init__(self, num_features=64, kernel_size_enc=2, hidden_dim=32, num_blocks=6, chunk_size=10, step_size=5,
                 bidir=True):
        super().__init__(DPTNConfig(num_features=num_features, hidden_video=num_features,
                                    kernel_size_enc=kernel_size_enc, hidden_dim=hidden_dim, num_blocks=num_blocks,
                                    chunk_size=chunk_size, step_size=step_size, bidir=bool(bidir), audio_only=True,
                                    arch="dprnn"))

    def forward(self, mix, **batch):
        return self._run(mix, None, None)


class DPRNNAVEncDec(_DPTNBase):
    """"DPRNN-AV" of BASELINE config 5.  The reference has no such class; this is
    DPRNNEncDec plus the lip-embedding fusion head of DPTNAVWavEncDec (dptn_wav.py:173-184) with the same parameter
    names (gate, visual_compression.*, video_ln.*)."""

    def __init__(self, num_features=64, video_emb_size=512, hidden_video=64, kernel_size_enc=2, hidden_dim=32,
                 num_blocks=6, chunk_size=10, step_size=5, bidir=True):
        super().__init__(DPTNConfig(num_features=num_features, video_emb_size=video_emb_size,
                                    hidden_video=hidden_video, kernel_size_enc=kernel_size_enc,
                                    hidden_dim=hidden_dim, num_blocks=num_blocks, chunk_size=chunk_size,
                                    step_size=step_size, bidir=bool(bidir), audio_only=False, arch="dprnn"))

    def forward(self, mix, s1_embedding, s2_embedding, **batch):
        return self._run(mix, s1_embedding, s2_embedding)
