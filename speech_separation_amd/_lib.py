"""ctypes binding of libdptnav.so (include/dptnav.h).  No torch types cross this boundary.

The library is looked up next to this file (built in-tree by ``__graft_entry__.build()`` /
``python -m speech_separation_amd.build``).  There is no CPU fallback: if the shared object is
missing or does not export the full ABI, importing callers get a RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPTNAV_LIB: another build of the library (same-box A/B of two builds: tools/train_ab.py, tools/ab_option.py)
LIB_PATH = os.environ.get("DPTNAV_LIB") or os.path.join(_HERE, "libdptnav.so")
ABI_VERSION = 3


class DptnavConfig(C.Structure):
    """struct dptnav_config (include/dptnav.h)."""

    _fields_ = [(n, C.c_int32) for n in (
        "num_features", "video_emb_size", "hidden_video", "kernel_size_enc", "hidden_dim", "num_blocks",
        "chunk_size", "step_size", "num_heads", "bidir", "audio_only", "arch")]


_vp, _fp, _i, _i64, _sz = C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_size_t

#: name -> (restype, argtypes): every symbol include/dptnav.h declares
SYMBOLS = {
    "dptnav_abi_version": (_i, []),
    "dptnav_create": (_i, [C.POINTER(DptnavConfig), C.POINTER(_vp)]),
    "dptnav_destroy": (None, [_vp]),
    "dptnav_last_error": (C.c_char_p, [_vp]),
    "dptnav_num_weights": (_i, [_vp]),
    "dptnav_weight_name": (C.c_char_p, [_vp, _i]),
    "dptnav_weight_numel": (_i64, [_vp, _i]),
    "dptnav_bind_weights": (_i, [_vp, C.POINTER(_fp), _i]),
    "dptnav_frames": (_i64, [_vp, _i64]),
    "dptnav_chunks": (_i64, [_vp, _i64]),
    "dptnav_workspace_bytes": (_sz, [_vp, _i, _i64, _i]),
    "dptnav_forward": (_i, [_vp, _fp, _fp, _fp, _i, _i64, _i, _fp, _fp, _vp, _sz, _vp]),
    "dptnav_stage_head": (_i, [_vp, _fp, _fp, _fp, _i, _i64, _i, _fp, _fp, _vp, _sz, _vp]),
    "dptnav_stage_path": (_i, [_vp, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _vp]),
    "dptnav_stage_tail": (_i, [_vp, _fp, _fp, _i, _i64, _fp, _fp, _vp, _sz, _vp]),
    "dptnav_sisnr_pairs": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _fp, _vp]),
    "dptnav_workspace_tap": (_i, [_vp, _i, _i64, _i, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "dptnav_bind_grads": (_i, [_vp, C.POINTER(_fp), _i]),
    "dptnav_train_path_tape_bytes": (_sz, [_vp, _i, _i]),
    "dptnav_train_bwd_workspace_bytes": (_sz, [_vp, _i, _i]),
    "dptnav_train_path_forward": (_i, [_vp, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "dptnav_train_path_backward": (_i, [_vp, _i, _i, _fp, _fp, _fp, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "dptnav_train_tape_bytes": (_sz, [_vp, _i, _i64, _i]),
    "dptnav_train_workspace_bytes": (_sz, [_vp, _i, _i64, _i]),
    "dptnav_train_forward": (_i, [_vp, _fp, _fp, _fp, _i, _i64, _i, _fp, _fp, _vp, _sz, _vp, _sz, _vp]),
    "dptnav_train_backward": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i, _i64, _i, _vp, _sz, _vp, _sz, _vp]),
    "dptnav_flat_offset": (_i64, [_vp, _i]),
    "dptnav_flat_numel": (_i64, [_vp]),
    "dptnav_tail_scratch_bytes": (_sz, [_vp, _i]),
    "dptnav_pit_sisnr_loss": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i64, C.c_float, _fp, _fp, _fp, _vp, _sz, _vp]),
    "dptnav_grad_clip": (_i, [_vp, _fp, _i64, C.c_float, _vp, _sz, _fp, _vp]),
    "dptnav_adamw_step": (_i, [_vp, _fp, _fp, _fp, _i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _vp]),
    "dptnav_dropout_mask": (_i, [_vp, _i, _i, _i, _i, _fp, _vp]),
    "dptnav_set_option": (_i, [_vp, C.c_char_p, _i]),
    "dptnav_profile_enable": (_i, [_vp, _i]),
    "dptnav_profile_collect": (_i, [_vp]),
    "dptnav_profile_reset": (_i, [_vp]),
    "dptnav_profile_num": (_i, []),
    "dptnav_profile_name": (C.c_char_p, [_i]),
    "dptnav_profile_ms": (C.c_double, [_vp, _i]),
    "dptnav_profile_count": (_i64, [_vp, _i]),
    "dptnav_flops_per_mixture": (C.c_double, [_vp, _i64]),
    "dptnav_min_bytes_per_mixture": (C.c_double, [_vp, _i64]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libdptnav.so and type every entry point; raises RuntimeError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; import it FIRST so that libdptnav resolves libamdhip64 to the
    # runtime torch already loaded -- device pointers and hipStream_t handles are only meaningful inside
    # one runtime instance.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` at the repo root).  speech_separation_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libdptnav.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.dptnav_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libdptnav ABI {lib.dptnav_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = lib
    return lib
