"""Host-side driver of libdptnav: owns the handle and the workspace tensor, hands raw device
pointers of torch tensors to the C ABI.  torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Tuple

import torch

from . import _lib
from .spec import DPTNConfig, state_dict_spec


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check(t: torch.Tensor, name: str, shape: Tuple[int, ...], device: torch.device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.device != device:
        raise ValueError(f"{name}: lives on {t.device}, engine is on {device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


class _DeviceBoundLib:
    """ctypes library proxy: calls every dptnav_* function inside `torch.cuda.device(device)`."""

    def __init__(self, lib, device):
        self._lib, self._device, self._cache = lib, device, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw, dev = getattr(self._lib, name), self._device

            def fn(*args, _raw=raw, _dev=dev):
                with torch.cuda.device(_dev):
                    return _raw(*args)
            self._cache[name] = fn
        return fn


class DptnEngine:
    """One handle <-> one device <-> the caller's current stream (include/dptnav.h threading contract)."""

    def __init__(self, cfg: DPTNConfig, device: torch.device | str = "cuda:0", alloc=None):
        """`alloc(nbytes) -> uint8 device tensor, 256-byte aligned`: where the engine takes every buffer it allocates itself
        from (workspaces, tapes, outputs, gradient buffers); default = the caching allocator.  The memory-safety tests
        pass an allocator that places each buffer against an unmapped page (tests/guardmem)."""
        self.cfg = cfg
        self.device = torch.device(device)
        self._alloc_hook = alloc
        self.options_set: Dict[str, int] = {}
        if self.device.type != "cuda":
            raise RuntimeError("DptnEngine needs a GPU device (PyTorch-ROCm 'cuda:N'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # every entry point runs with THIS engine's GPU current (the library creates its internal streams / events on the
        # current HIP device and launches on the caller's stream, which must belong to it): rank-per-GPU callers need
        # not have called torch.cuda.set_device themselves
        self.lib = _DeviceBoundLib(_lib.load(), self.device)
        c = _lib.DptnavConfig(cfg.num_features, cfg.video_emb_size, cfg.hidden_video, cfg.kernel_size_enc,
                              cfg.hidden_dim, cfg.num_blocks, cfg.chunk_size, cfg.step_size, cfg.num_heads,
                              int(cfg.bidir), int(cfg.audio_only), {"dptn": 0, "dprnn": 1}[cfg.arch])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.dptnav_create(C.byref(c), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"dptnav_create failed ({rc}): {self.lib.dptnav_last_error(None).decode()}")
        self._h = h
        self._ws: Optional[torch.Tensor] = None
        self._bound: Optional[list] = None  # keeps the tensors alive
        # the library's slot table must equal the Python spec (both restate the reference's state_dict)
        spec = state_dict_spec(cfg)
        n = self.lib.dptnav_num_weights(h)
        names = [self.lib.dptnav_weight_name(h, i).decode() for i in range(n)]
        if names != [k for k, _ in spec]:
            raise RuntimeError("libdptnav weight table disagrees with speech_separation_amd.spec")
        self.slots = spec

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dptnav_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, rc: int, what: str):
        raise RuntimeError(f"{what} failed ({rc}): {self.lib.dptnav_last_error(self._h).decode()}")

    # ------------------------------------------------------------------ device memory
    def _alloc(self, nbytes: int) -> torch.Tensor:
        if self._alloc_hook is None:
            return torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        t = self._alloc_hook(int(nbytes))
        if t.dtype != torch.uint8 or t.device != self.device or t.numel() < nbytes or (nbytes and t.data_ptr() % 256):
            raise ValueError("alloc hook must return a 256-byte aligned uint8 tensor of the requested size on the engine's device")
        return t

    def _empty(self, *shape) -> torch.Tensor:
        if self._alloc_hook is None:
            return torch.empty(*shape, device=self.device)
        n = 1
        for d in shape:
            n *= int(d)
        return self._alloc(4 * n)[:4 * n].view(torch.float32).view(*shape)

    def _check_ws(self, ws: torch.Tensor, need: int, name: str = "ws") -> torch.Tensor:
        """A caller-provided workspace: uint8, on this engine's device, 256-byte aligned, at least `need` bytes."""
        if not isinstance(ws, torch.Tensor) or ws.dtype != torch.uint8:
            raise TypeError(f"{name}: expected a uint8 tensor")
        if ws.device != self.device:
            raise ValueError(f"{name}: lives on {ws.device}, engine is on {self.device}")
        if not ws.is_contiguous() or ws.data_ptr() % 256:
            raise ValueError(f"{name}: must be contiguous and 256-byte aligned")
        if ws.numel() < need:
            raise ValueError(f"{name}: {ws.numel()} bytes, this call needs {need}")
        return ws

    # ------------------------------------------------------------------ weights
    def bind(self, params: Mapping[str, torch.Tensor]):
        """Borrow the parameter storages (no copies): call again if they are re-allocated."""
        keep, ptrs = [], (C.c_void_p * len(self.slots))()
        for i, (key, shape) in enumerate(self.slots):
            if key not in params:
                raise KeyError(f"missing parameter {key}")
            t = params[key].detach()
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{key}: expected {tuple(shape)}, got {tuple(t.shape)}")
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{key}: must be contiguous float32 on {self.device}")
            keep.append(t)
            ptrs[i] = t.data_ptr()
        rc = self.lib.dptnav_bind_weights(self._h, ptrs, len(self.slots))
        if rc:
            self._raise(rc, "dptnav_bind_weights")
        self._bound = keep
        self._bound_ptrs = tuple(t.data_ptr() for t in keep)

    def bound_to(self, params: Mapping[str, torch.Tensor]) -> bool:
        if self._bound is None:
            return False
        return self._bound_ptrs == tuple(params[k].data_ptr() for k, _ in self.slots)

    # ------------------------------------------------------------------ sizes / workspace
    def frames(self, T: int) -> int:
        return int(self.lib.dptnav_frames(self._h, T))

    def chunks(self, T: int) -> int:
        return int(self.lib.dptnav_chunks(self._h, T))

    def workspace_bytes(self, B: int, T: int, Tv: int) -> int:
        n = int(self.lib.dptnav_workspace_bytes(self._h, B, T, Tv))
        if n == 0:
            raise RuntimeError(f"unsupported shape: {self.lib.dptnav_last_error(self._h).decode()}")
        return n

    def _workspace(self, B: int, T: int, Tv: int) -> torch.Tensor:
        need = self.workspace_bytes(B, T, Tv)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = self._alloc(need)
        assert self._ws.data_ptr() % 256 == 0
        return self._ws

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def flops_per_mixture(self, T: int) -> float:
        return float(self.lib.dptnav_flops_per_mixture(self._h, T))

    def min_bytes_per_mixture(self, T: int) -> float:
        return float(self.lib.dptnav_min_bytes_per_mixture(self._h, T))

    def set_option(self, key: str, value: int):
        rc = self.lib.dptnav_set_option(self._h, key.encode(), int(value))
        if rc:
            self._raise(rc, "dptnav_set_option")
        self.options_set[key] = int(value)        # what this wrapper has set (the library's defaults are in dptnav.h)

    # ------------------------------------------------------------------ per-kernel device timing
    def profile(self, on: bool):
        self.lib.dptnav_profile_enable(self._h, int(on))

    def profile_reset(self):
        rc = self.lib.dptnav_profile_reset(self._h)
        if rc:
            self._raise(rc, "dptnav_profile_reset")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        """{kernel class: (total device ms, launches)} since the last reset (waits for recorded events)."""
        rc = self.lib.dptnav_profile_collect(self._h)
        if rc:
            self._raise(rc, "dptnav_profile_collect")
        out = {}
        for i in range(self.lib.dptnav_profile_num()):
            out[self.lib.dptnav_profile_name(i).decode()] = (float(self.lib.dptnav_profile_ms(self._h, i)),
                                                            int(self.lib.dptnav_profile_count(self._h, i)))
        return out

    # ------------------------------------------------------------------ hot path
    def forward(self, mix: torch.Tensor, e1: Optional[torch.Tensor] = None, e2: Optional[torch.Tensor] = None,
                out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                ws: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        cfg = self.cfg
        if mix.dim() != 2:
            raise ValueError(f"mix: expected (B,T), got {tuple(mix.shape)}")
        B, T = mix.shape
        mix = _check(mix, "mix", (B, T), self.device)
        Tv = 1
        if not cfg.audio_only:
            if e1 is None or e2 is None:
                raise ValueError("s1_embedding and s2_embedding are required for the audio-visual model")
            Tv = e1.shape[-1]
            e1 = _check(e1, "s1_embedding", (B, cfg.video_emb_size, Tv), self.device)
            e2 = _check(e2, "s2_embedding", (B, cfg.video_emb_size, Tv), self.device)
        else:
            e1 = e2 = None
        if ws is None:
            ws = self._workspace(B, T, Tv)
        else:
            ws = self._check_ws(ws, self.workspace_bytes(B, T, Tv))
        if out is not None:
            s1, s2 = (_check(t, n, (B, T), self.device) for t, n in zip(out, ("out[0]", "out[1]")))
            if s1.data_ptr() != out[0].data_ptr() or s2.data_ptr() != out[1].data_ptr():
                raise ValueError("out: the output tensors must be contiguous")
        else:
            s1, s2 = self._empty(B, T), self._empty(B, T)
        rc = self.lib.dptnav_forward(self._h, mix.data_ptr(), _ptr(e1), _ptr(e2), B, T, Tv, s1.data_ptr(),
                                     s2.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_forward")
        return s1, s2

    # ------------------------------------------------------------------ training step, path level
    def bind_grads(self, per_slot: bool = False) -> Dict[str, torch.Tensor]:
        """Allocate the gradient buffers the library WRITES and bind them; returns {key: tensor}.  All of them are views
        of ONE flat tensor (`self._grads_flat`, every slot at a 256-byte aligned offset `self._grad_offsets[key]`), so
        that a step can be handed to autograd with one copy and to RCCL with one collective.  per_slot = True (tests):
        every slot is an allocation of its own, of exactly its size (no flat tensor; the C ABI takes any pointers)."""
        offs = self.flat_offsets()
        flat = None if per_slot else self._empty(self.flat_numel()).zero_()
        grads, ptrs = {}, (C.c_void_p * len(self.slots))()
        for i, (key, shape) in enumerate(self.slots):
            n = 1
            for d in shape:
                n *= int(d)
            g = self._empty(*shape).zero_() if per_slot else flat[offs[key]:offs[key] + n].view(*shape)
            grads[key] = g
            ptrs[i] = g.data_ptr()
        rc = self.lib.dptnav_bind_grads(self._h, ptrs, len(self.slots))
        if rc:
            self._raise(rc, "dptnav_bind_grads")
        self._grads, self._grads_flat, self._grad_offsets = grads, flat, offs
        return grads

    # ------------------------------------------------------------------ flat layout / training tail (N1)
    def flat_offsets(self) -> Dict[str, int]:
        """{state_dict key: offset in floats} of the flat gradient / optimizer-state layout (include/dptnav.h)."""
        return {key: int(self.lib.dptnav_flat_offset(self._h, i)) for i, (key, _) in enumerate(self.slots)}

    def flat_numel(self) -> int:
        return int(self.lib.dptnav_flat_numel(self._h))

    def _tail_scratch(self, B: int) -> torch.Tensor:
        need = int(self.lib.dptnav_tail_scratch_bytes(self._h, max(B, 1)))
        if getattr(self, "_tail_ws", None) is None or self._tail_ws.numel() < need:
            self._tail_ws = self._alloc(need)
        return self._tail_ws

    def pit_sisnr_loss(self, s1_pred, s2_pred, s1, s2, grad_scale: float = 1.0):
        """-> (d loss / d s1_pred, d loss / d s2_pred, out[4] = loss, permutation, loss perm 0, loss perm 1); all on the
        device, no synchronisation."""
        B, T = s1_pred.shape
        ts = [_check(t.detach(), n, (B, T), self.device) for t, n in ((s1_pred, "s1_pred"), (s2_pred, "s2_pred"), (s1, "s1"),
                                                                     (s2, "s2"))]
        d1, d2 = self._empty(B, T), self._empty(B, T)
        out = self._empty(4)
        ws = self._tail_scratch(B)
        rc = self.lib.dptnav_pit_sisnr_loss(self._h, *[t.data_ptr() for t in ts], B, T, float(grad_scale), d1.data_ptr(),
                                            d2.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_pit_sisnr_loss")
        return d1, d2, out

    def grad_clip(self, flat_grad: torch.Tensor, max_norm: Optional[float]) -> torch.Tensor:
        """Scales `flat_grad` (flat layout) in place like clip_grad_norm_; returns the pre-clip norm as a 0-dim device tensor."""
        flat_grad = _check(flat_grad, "flat_grad", (self.flat_numel(),), self.device)
        norm = self._empty(1)
        ws = self._tail_scratch(1)
        rc = self.lib.dptnav_grad_clip(self._h, flat_grad.data_ptr(), flat_grad.numel(),
                                       float(max_norm) if max_norm is not None else 0.0, ws.data_ptr(), ws.numel(),
                                       norm.data_ptr(), self._stream())
        if rc:
            self._raise(rc, "dptnav_grad_clip")
        return norm[0]

    def adamw_step(self, flat_grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step: int):
        n = self.flat_numel()
        for t, name in ((flat_grad, "flat_grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            _check(t, name, (n,), self.device)
            if not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous")
        rc = self.lib.dptnav_adamw_step(self._h, flat_grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n, float(lr),
                                        float(beta1), float(beta2), float(eps), float(weight_decay), int(step), self._stream())
        if rc:
            self._raise(rc, "dptnav_adamw_step")

    def train_path_forward(self, block: int, path: int, x: torch.Tensor):
        B, S, K, N = x.shape
        x = _check(x, "x", (B, S, self.cfg.chunk_size, self.cfg.num_features), self.device)
        ws = self._workspace(B, self._path_T(S), 1)
        tape = self._alloc(int(self.lib.dptnav_train_path_tape_bytes(self._h, B, S)))
        y = self._empty(*x.shape)
        rc = self.lib.dptnav_train_path_forward(self._h, block, path, x.data_ptr(), y.data_ptr(), B, S, tape.data_ptr(),
                                                tape.numel(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_train_path_forward")
        return y, tape

    def train_path_backward(self, block: int, path: int, x: torch.Tensor, dy: torch.Tensor, tape: torch.Tensor):
        B, S, K, N = x.shape
        dy = _check(dy, "dy", tuple(x.shape), self.device)
        need = int(self.lib.dptnav_train_bwd_workspace_bytes(self._h, B, S))
        if getattr(self, "_bws", None) is None or self._bws.numel() < need:
            self._bws = None
            self._bws = self._alloc(need)
        dx = self._empty(*x.shape)
        rc = self.lib.dptnav_train_path_backward(self._h, block, path, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, S,
                                                 tape.data_ptr(), tape.numel(), self._bws.data_ptr(), self._bws.numel(),
                                                 self._stream())
        if rc:
            self._raise(rc, "dptnav_train_path_backward")
        return dx

    def dropout_mask(self, block: int, path: int, B: int, S: int) -> torch.Tensor:
        """(nseq, heads, len, len) keep-mask of path (block, path) under the current dropout options (test helper)."""
        K = self.cfg.chunk_size
        nseq, ln = (B * S, K) if path == 0 else (B * K, S)
        m = self._empty(nseq, self.cfg.num_heads, ln, ln)
        rc = self.lib.dptnav_dropout_mask(self._h, block, path, B, S, m.data_ptr(), self._stream())
        if rc:
            self._raise(rc, "dptnav_dropout_mask")
        return m

    # ------------------------------------------------------------------ training step, whole model
    def train_forward(self, mix, e1=None, e2=None):
        """Forward that records the tape; returns (s1_pred, s2_pred, tape)."""
        B, T = mix.shape
        mix = _check(mix, "mix", (B, T), self.device)
        Tv = 1 if self.cfg.audio_only else e1.shape[-1]
        if not self.cfg.audio_only:
            e1 = _check(e1, "s1_embedding", (B, self.cfg.video_emb_size, Tv), self.device)
            e2 = _check(e2, "s2_embedding", (B, self.cfg.video_emb_size, Tv), self.device)
        nt = int(self.lib.dptnav_train_tape_bytes(self._h, B, T, Tv))
        nw = int(self.lib.dptnav_train_workspace_bytes(self._h, B, T, Tv))
        if nt == 0 or nw == 0:
            raise RuntimeError(f"training step unsupported: {self.lib.dptnav_last_error(self._h).decode()}")
        tape = self._alloc(nt)
        if getattr(self, "_tws", None) is None or self._tws.numel() < nw:
            self._tws = None
            self._tws = self._alloc(nw)
        s1, s2 = self._empty(B, T), self._empty(B, T)
        rc = self.lib.dptnav_train_forward(self._h, mix.data_ptr(), _ptr(e1), _ptr(e2), B, T, Tv, s1.data_ptr(), s2.data_ptr(),
                                           tape.data_ptr(), tape.numel(), self._tws.data_ptr(), self._tws.numel(),
                                           self._stream())
        if rc:
            self._raise(rc, "dptnav_train_forward")
        return s1, s2, tape

    def train_backward(self, mix, e1, e2, d_s1, d_s2, tape):
        """Writes every parameter gradient into the buffers of bind_grads()."""
        B, T = mix.shape
        Tv = 1 if self.cfg.audio_only else e1.shape[-1]
        d_s1 = _check(d_s1, "d_s1_pred", (B, T), self.device)
        d_s2 = _check(d_s2, "d_s2_pred", (B, T), self.device)
        rc = self.lib.dptnav_train_backward(self._h, mix.data_ptr(), _ptr(e1), _ptr(e2), d_s1.data_ptr(), d_s2.data_ptr(), B,
                                            T, Tv, tape.data_ptr(), tape.numel(), self._tws.data_ptr(), self._tws.numel(),
                                            self._stream())
        if rc:
            self._raise(rc, "dptnav_train_backward")

    # ------------------------------------------------------------------ loss / metric statistics
    def sisnr_pairs(self, s1_pred, s2_pred, s1, s2, mix) -> torch.Tensor:
        """(B,6,2) device tensor: [metric dB, loss term] for the pairs (p1,s1) (p1,s2) (p2,s1) (p2,s2) (mix,s1) (mix,s2)."""
        B, T = mix.shape
        ts = [_check(t, n, (B, T), self.device) for t, n in ((s1_pred, "s1_pred"), (s2_pred, "s2_pred"), (s1, "s1"),
                                                             (s2, "s2"), (mix, "mix"))]
        out = self._empty(B, 6, 2)
        rc = self.lib.dptnav_sisnr_pairs(self._h, *[t.data_ptr() for t in ts], B, T, out.data_ptr(), self._stream())
        if rc:
            self._raise(rc, "dptnav_sisnr_pairs")
        return out

    # ------------------------------------------------------------------ stages (parity tests / profiling)
    def stage_head(self, mix, e1=None, e2=None):
        cfg = self.cfg
        B, T = mix.shape
        Tv = 1 if cfg.audio_only else e1.shape[-1]
        L, S = self.frames(T), self.chunks(T)
        ws = self._workspace(B, T, Tv)
        enc = self._empty(B, L, cfg.num_features)
        chk = self._empty(B, S, cfg.chunk_size, cfg.num_features)
        rc = self.lib.dptnav_stage_head(self._h, mix.contiguous().data_ptr(),
                                        _ptr(None if e1 is None else e1.contiguous()),
                                        _ptr(None if e2 is None else e2.contiguous()), B, T, Tv, enc.data_ptr(),
                                        chk.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_stage_head")
        return enc, chk

    def _path_T(self, S: int) -> int:
        L = (S - 1) * self.cfg.step_size + self.cfg.chunk_size
        return (L - 1) * self.cfg.stride_enc + self.cfg.kernel_size_enc

    def stage_path(self, block: int, path: int, x: torch.Tensor) -> torch.Tensor:
        B, S, K, N = x.shape
        x = _check(x, "x", (B, S, self.cfg.chunk_size, self.cfg.num_features), self.device)
        ws = self._workspace(B, self._path_T(S), 1)
        y = self._empty(*x.shape)
        rc = self.lib.dptnav_stage_path(self._h, block, path, x.data_ptr(), y.data_ptr(), B, S, ws.data_ptr(),
                                        ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_stage_path")
        return y

    def stage_tail(self, x: torch.Tensor, encoded: torch.Tensor, T: int):
        B = x.shape[0]
        ws = self._workspace(B, T, 1)
        s1 = self._empty(B, T)
        s2 = self._empty(B, T)
        rc = self.lib.dptnav_stage_tail(self._h, x.contiguous().data_ptr(), encoded.contiguous().data_ptr(), B, T,
                                        s1.data_ptr(), s2.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            self._raise(rc, "dptnav_stage_tail")
        return s1, s2

    def tap(self, name: str, B: int, T: int, Tv: int = 1) -> torch.Tensor:
        """View of an intermediate left in the workspace by the last stage_path/forward call."""
        off, n = C.c_size_t(), C.c_size_t()
        rc = self.lib.dptnav_workspace_tap(self._h, B, T, Tv, name.encode(), C.byref(off), C.byref(n))
        if rc:
            self._raise(rc, "dptnav_workspace_tap")
        ws = self._workspace(B, T, Tv)
        return ws[off.value:off.value + 4 * n.value].view(torch.float32)


def params_to_device(sd: Mapping[str, "object"], device) -> Dict[str, torch.Tensor]:
    """numpy / torch state_dict -> contiguous float32 device tensors."""
    out = {}
    for k, v in sd.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
        out[k] = t.to(device=device, dtype=torch.float32).contiguous()
    return out
