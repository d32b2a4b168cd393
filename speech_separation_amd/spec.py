"""Shape / checkpoint specification of the DPTN(-AV) raw-waveform separator.

This module is pure host logic (numpy only): it states the hyper-parameters the
hot path is built for, the derived sizes (frames, chunks, tokens) and the exact
``state_dict`` key order + shapes that the reference's checkpoints use, so that
the HIP library, the ``nn.Module`` shim, the oracle and the tests all agree on
one table.

Reference surfaces restated here (never imported):
  * ctor kwargs / defaults ........ src/model/dptn_wav.py:137-169 (AV), :72-103 (audio)
  * Hydra values .................. src/configs/model/dptn_wav_av.yaml:1-12,
                                    src/configs/model/dptn_wav.yaml:1-10
  * parameter registration order .. dptn_wav.py:153-169 -> dptn_wav.py:18-33
                                    -> dptn.py:59-60 -> dptn.py:16-34
  * frame/chunk arithmetic ........ nn.Conv1d(k, stride=k//2) dptn_wav.py:153;
                                    SplitToFolds dprnn.py:122-136
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np


@dataclass(frozen=True)
class DPTNConfig:
    """Constructor arguments of DPTNAVWavEncDec / DPTNWavEncDec (same names)."""

    num_features: int = 128
    video_emb_size: int = 512
    hidden_video: int = 128
    kernel_size_enc: int = 7
    hidden_dim: int = 128
    num_blocks: int = 6
    chunk_size: int = 150
    step_size: int = 75
    num_heads: int = 4
    dropout: float = 0.1
    bidir: bool = True
    audio_only: bool = False  # True -> DPTNWavEncDec / DPRNNEncDec (no video branch)
    arch: str = "dptn"        # "dptn": MHA + bi-LSTM blocks (dptn.py); "dprnn": bi-LSTM + fc blocks (dprnn.py:7-113)

    # ---- derived sizes -------------------------------------------------
    @property
    def stride_enc(self) -> int:
        return self.kernel_size_enc // 2

    @property
    def head_dim(self) -> int:
        return self.num_features // self.num_heads

    def frames(self, T: int) -> int:
        """Latent length L of Conv1d(k, stride=k//2, no padding)."""
        return (T - self.kernel_size_enc) // self.stride_enc + 1

    def chunks(self, L: int) -> int:
        """Number of 50%-overlap chunks S produced by SplitToFolds."""
        return (L - self.chunk_size) // self.step_size + 1

    def ola_len(self, S: int) -> int:
        return (S - 1) * self.step_size + self.chunk_size

    def tokens(self, B: int, T: int) -> int:
        return B * self.chunks(self.frames(T)) * self.chunk_size

    def to_dict(self) -> dict:
        return asdict(self)


#: BASELINE.json config 3/4 (src/configs/model/dptn_wav_av.yaml)
DPTN_AV = DPTNConfig()
#: BASELINE.json config 2 (src/configs/model/dptn_wav.yaml)
DPTN_AUDIO = DPTNConfig(num_features=64, audio_only=True)
#: BASELINE.json config 5 backbone (src/configs/model/dprnn.yaml): audio-only DPRNNEncDec is the only variant the
#: reference defines; audio_only=False adds this repo's AV fusion head (same as dptn_wav.py:173-184) = "DPRNN-AV"
DPRNN_AUDIO = DPTNConfig(num_features=64, hidden_video=64, kernel_size_enc=2, hidden_dim=128, num_blocks=6,
                         chunk_size=250, step_size=125, audio_only=True, arch="dprnn")
DPRNN_AV = DPTNConfig(**{**DPRNN_AUDIO.__dict__, "audio_only": False})
#: small shape used by the golden fixtures (fast on every backend)
DPTN_TINY = DPTNConfig(num_features=32, video_emb_size=24, hidden_video=32, kernel_size_enc=7,
                       hidden_dim=32, num_blocks=2, chunk_size=10, step_size=5, num_heads=4)


_DPRNN_PATH_TENSORS: List[Tuple[str, str]] = [
    ("rnn.weight_ih_l0", "4H,N"),
    ("rnn.weight_hh_l0", "4H,H"),
    ("rnn.bias_ih_l0", "4H"),
    ("rnn.bias_hh_l0", "4H"),
    ("rnn.weight_ih_l0_reverse", "4H,N"),
    ("rnn.weight_hh_l0_reverse", "4H,H"),
    ("rnn.bias_ih_l0_reverse", "4H"),
    ("rnn.bias_hh_l0_reverse", "4H"),
    ("fc.weight", "N,DH"),
    ("fc.bias", "N"),
    ("norm1d.weight", "N"),
    ("norm1d.bias", "N"),
]

_PATH_TENSORS: List[Tuple[str, str]] = [
    # (suffix, shape-code)
    ("mha.in_proj_weight", "3N,N"),
    ("mha.in_proj_bias", "3N"),
    ("mha.out_proj.weight", "N,N"),
    ("mha.out_proj.bias", "N"),
    ("ln1.weight", "N"),
    ("ln1.bias", "N"),
    ("rnn.weight_ih_l0", "4H,N"),
    ("rnn.weight_hh_l0", "4H,H"),
    ("rnn.bias_ih_l0", "4H"),
    ("rnn.bias_hh_l0", "4H"),
    ("rnn.weight_ih_l0_reverse", "4H,N"),
    ("rnn.weight_hh_l0_reverse", "4H,H"),
    ("rnn.bias_ih_l0_reverse", "4H"),
    ("rnn.bias_hh_l0_reverse", "4H"),
    ("ffn.1.weight", "N,DH"),
    ("ffn.1.bias", "N"),
    ("ln2.weight", "N"),
    ("ln2.bias", "N"),
]


def state_dict_spec(cfg: DPTNConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list == ``model.state_dict()`` of the reference.

    The order is the slot order of the C-ABI weight table (include/dptnav.h).
    """
    N, H, k = cfg.num_features, cfg.hidden_dim, cfg.kernel_size_enc
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if not cfg.audio_only:
        out.append(("gate", (1,)))
    out.append(("encoder.weight", (N, 1, k)))
    if not cfg.audio_only:
        out.append(("visual_compression.weight", (cfg.hidden_video // 2, cfg.video_emb_size)))
        out.append(("visual_compression.bias", (cfg.hidden_video // 2,)))
        out.append(("video_ln.weight", (cfg.hidden_video,)))
        out.append(("video_ln.bias", (cfg.hidden_video,)))
    for b in range(cfg.num_blocks):
        for path in ("intra_chunk_block", "inter_chunk_block"):
            two_dirs = True if path == "intra_chunk_block" else cfg.bidir
            dims = {"N": N, "3N": 3 * N, "4H": 4 * H, "H": H, "DH": H * (2 if two_dirs else 1)}
            for suffix, code in (_PATH_TENSORS if cfg.arch == "dptn" else _DPRNN_PATH_TENSORS):
                if suffix.endswith("_reverse") and not two_dirs:
                    continue
                shape = tuple(dims[c] for c in code.split(","))
                out.append((f"dprnn.model.{b}.{path}.{suffix}", shape))
    out.append(("dprnn.speakers_separation.0.weight", (1,)))
    out.append(("dprnn.speakers_separation.1.weight", (2 * N, N, 1, 1)))
    out.append(("dprnn.speakers_separation.1.bias", (2 * N,)))
    out.append(("dprnn.postprocessing.0.weight", (N, N, 1)))
    out.append(("dprnn.postprocessing.0.bias", (N,)))
    out.append(("decoder.weight", (N, 1, k)))
    return out


def num_parameters(cfg: DPTNConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in state_dict_spec(cfg)))


def synthetic_state_dict(cfg: DPTNConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic, platform-independent random weights (numpy PCG64).

    Used wherever the same weights must exist on two machines without shipping
    them (golden generation here, parity tests / bench on the GPU box).  The
    distributions mimic torch's default init scale (uniform +-1/sqrt(fan_in))
    but make every tensor non-trivial (LayerNorm gains != 1, biases != 0) so a
    dropped parameter shows up in parity.
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for key, shape in state_dict_spec(cfg):
        leaf = key.split(".")[-1]
        if key == "gate":
            w = np.array([0.7], dtype=np.float64)
        elif key == "dprnn.speakers_separation.0.weight":
            w = np.array([0.25], dtype=np.float64)
        elif (key.split(".")[-2].startswith(("ln", "video_ln", "norm1d"))) and leaf == "weight":
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif (key.split(".")[-2].startswith(("ln", "video_ln", "norm1d"))) and leaf == "bias":
            w = 0.05 * rng.standard_normal(shape)
        else:
            if leaf.startswith("bias") or leaf.endswith("bias"):
                fan_in = {
                    "mha.in_proj_bias": cfg.num_features,
                }.get(".".join(key.split(".")[-2:]), cfg.hidden_dim)
            else:
                fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            bound = 1.0 / np.sqrt(max(fan_in, 1))
            w = rng.uniform(-bound, bound, size=shape)
        sd[key] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


def synthetic_inputs(cfg: DPTNConfig, B: int, T: int, Tv: int = 50, seed: int = 123) -> Dict[str, np.ndarray]:
    """Seeded inputs of the shapes the trainer hands to ``model(**batch)``.

    mix (B,T) comes from src/datasets/base_dataset.py:144-145 + collate.py:43,
    s*_embedding (B,512,50) from make_embeddings.py:65-69 / base_dataset.py:199-205.
    s1/s2 are only used by loss/metric checks; mix = s1 + s2.
    """
    rng = np.random.default_rng(seed)
    s1 = (0.1 * rng.standard_normal((B, T))).astype(np.float32)
    s2 = (0.1 * rng.standard_normal((B, T))).astype(np.float32)
    out = {"s1": s1, "s2": s2, "mix": (s1 + s2).astype(np.float32)}
    if not cfg.audio_only:
        out["s1_embedding"] = rng.standard_normal((B, cfg.video_emb_size, Tv)).astype(np.float32)
        out["s2_embedding"] = rng.standard_normal((B, cfg.video_emb_size, Tv)).astype(np.float32)
    return out
