"""speech_separation_amd -- MI355X-native DPTN(-AV) separation forward path (libdptnav + thin host layer)."""
from .spec import DPRNN_AUDIO, DPRNN_AV, DPTN_AUDIO, DPTN_AV, DPTNConfig, state_dict_spec, synthetic_inputs, synthetic_state_dict

__all__ = ["DPTNConfig", "DPTN_AV", "DPTN_AUDIO", "DPRNN_AUDIO", "DPRNN_AV", "DPRNNEncDec", "DPRNNAVEncDec", "state_dict_spec", "synthetic_state_dict", "synthetic_inputs",
           "DptnEngine", "DPTNAVWavEncDec", "DPTNWavEncDec", "FusedAdamW", "clip_grad_norm_", "SiSNRWavLoss"]


def __getattr__(name):  # torch-dependent parts are imported lazily (spec.py stays numpy-only)
    if name == "DptnEngine":
        from .engine import DptnEngine
        return DptnEngine
    if name in ("DPTNAVWavEncDec", "DPTNWavEncDec", "DPRNNEncDec", "DPRNNAVEncDec"):
        from . import model
        return getattr(model, name)
    if name in ("FusedAdamW", "clip_grad_norm_"):
        from . import optim
        return getattr(optim, name)
    if name == "SiSNRWavLoss":
        from .metrics import SiSNRWavLoss
        return SiSNRWavLoss
    raise AttributeError(name)
