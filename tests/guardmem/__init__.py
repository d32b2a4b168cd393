"""TEST INFRASTRUCTURE: guard-page device buffers for the memory-safety harness (tests/test_gpu_memsafety.py).

`GuardArena.bytes(n)` / `.floats(shape)` return torch tensors that alias HIP virtual-memory allocations with an unmapped
granule on either side (guardmem.cpp): a kernel that reads or writes one element past such a buffer raises a GPU memory
access fault instead of silently touching a neighbouring tensor of the caching allocator.  The product never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "guardmem.cpp")
OUT = os.path.join(HERE, "libguardmem.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force: bool = False) -> str:
    """Host-only C++ against the HIP runtime headers; compiles without a GPU."""
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        r = subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", SRC, "-o", OUT],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"guardmem build failed:\n{r.stderr[-2000:]}")
    return OUT


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.gm_last_error.restype = C.c_char_p
        lib.gm_granularity.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        lib.gm_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        lib.gm_mapped_range.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.gm_free.argtypes = [C.c_void_p]
        lib.gm_chunks.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


class _Raw:
    """What torch.as_tensor aliases: an object with __cuda_array_interface__ (the tensor keeps it alive)."""

    def __init__(self, ptr: int, nbytes: int, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


class GuardArena:
    """All guard allocations of one test; `flush` = "end" (overruns fault) or "start" (underruns fault)."""

    def __init__(self, device_index: int = 0, flush: str = "end", fill: int = 0xFF):
        import torch
        assert flush in ("end", "start")
        self.lib, self.dev, self.flush, self.fill = load(), device_index, flush, fill
        self.handles, self.total = [], 0
        self._torch = torch
        g = C.c_size_t()
        if self.lib.gm_granularity(self.dev, C.byref(g)):
            raise RuntimeError(self.lib.gm_last_error().decode())
        self.granularity = g.value

    def bytes(self, nbytes: int, align: int = 256):
        """uint8 tensor of nbytes on guard pages; its whole mapping is pre-filled with `fill` (0xFF.. = NaN as fp32)."""
        torch = self._torch
        nbytes = int(nbytes)
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=f"cuda:{self.dev}")
        p, h = C.c_void_p(), C.c_void_p()
        if self.lib.gm_alloc(self.dev, nbytes, align, int(self.flush == "end"), C.byref(p), C.byref(h)):
            raise RuntimeError(self.lib.gm_last_error().decode())
        self.handles.append(h)
        self.total += nbytes
        base, mapped = C.c_void_p(), C.c_size_t()
        self.lib.gm_mapped_range(h, C.byref(base), C.byref(mapped))
        if os.environ.get("GUARDMEM_LOG"):      # one line per allocation: a fault address can be looked up afterwards
            with open(os.environ["GUARDMEM_LOG"], "a") as fh:
                fh.write(f"{len(self.handles)} flush={self.flush} bytes={nbytes} user=0x{p.value:x} mapped=[0x{base.value:x}, 0x{base.value + mapped.value:x})\n")
        whole = torch.as_tensor(_Raw(base.value, mapped.value, self), device=f"cuda:{self.dev}")
        whole.fill_(self.fill)
        t = torch.as_tensor(_Raw(p.value, nbytes, self), device=f"cuda:{self.dev}")
        assert t.data_ptr() == p.value and t.numel() == nbytes
        if self.flush == "end":
            assert base.value + mapped.value - (p.value + nbytes) < align
        else:
            assert p.value == base.value
        return t

    def floats(self, *shape, align: int = 16):
        n = 1
        for d in shape:
            n *= int(d)
        return self.bytes(4 * n, align).view(self._torch.float32).view(*shape)

    def like(self, t, align: int = 16):
        """A guard-page copy of a float32 / uint8 device or host tensor."""
        torch = self._torch
        if t.dtype == torch.uint8:
            g = self.bytes(t.numel(), align).view(t.shape)
        else:
            assert t.dtype == torch.float32
            g = self.floats(*t.shape, align=align) if t.numel() else torch.empty_like(t, device=f"cuda:{self.dev}")
        if t.numel():
            g.copy_(t)
        return g

    def chunks(self):
        """physical handles per allocation, in allocation order (a buffer above 1 GiB is backed by several)"""
        return [int(self.lib.gm_chunks(h)) for h in self.handles]

    def close(self):
        """Unmap, release and free every allocation; every step's return code is checked (gm_free reports the first that failed)."""
        self._torch.cuda.synchronize(self.dev)
        errors = []
        for h in self.handles:
            if self.lib.gm_free(h):
                errors.append(self.lib.gm_last_error().decode())
        self.handles = []
        if errors:
            raise RuntimeError(f"guardmem: {len(errors)} allocation(s) did not come apart cleanly, first: {errors[0]}")
