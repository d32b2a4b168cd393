"""In-tree build of libdptnav.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m speech_separation_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdptnav.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# translation units and their extra flags.  lstm.hip / lstm16.hip: MFMA accumulators in architectural VGPRs so that the 256 W_hh
# fragments own the AGPRs and the step loop carries no v_accvgpr moves (see the file's header).
# dptnav.hip: the GEMM engine's tile-ticket atomicAdd is issued by ONE lane a whole MFMA block before its result is used;
# LLVM's atomic optimizer would turn it into a wave reduction + broadcast that waits for the result on the spot (wave 0
# then sits out the round trip of a contended atomic every tile, the other waves wait for it at the next barrier).
SOURCES = {"dptnav.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"], "lstm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "lstm16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "lstm_bptt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "lstm_bptt16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           # attn_block.hip: the softmax works on MFMA results with plain VALU instructions -> accumulators in architectural VGPRs
           "attn_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "attn_block2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "dgrad_t.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "gemm_t.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           # dgrad_r.hip: 160 accumulator registers (data gradient + the riding weight gradient) -> AGPR form, weights in VGPRs
           "dgrad_r.hip": [],
           "attn_block64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "lstm16s.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "lstm4.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
           "lstm16x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "fcln.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _headers():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "dptnav.h")]


def source_digest() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: profiles record it, bench.py compares
    it with the tree it runs from and says so when a committed PMC table was taken with other kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(list(SOURCES) + _headers()):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in list(SOURCES) + _headers())


def build_lib(force: bool = False, verbose: bool = False, out: str = OUT, extra_flags=()) -> str:
    """Compile csrc/dptnav.hip.  `out` / `extra_flags` exist for A/B experiments (tools/): the product is OUT."""
    if out == OUT and not force and not _stale():
        return OUT
    log = os.path.join(HERE, "csrc", "build_resource_usage.log")
    objdir = os.path.join(HERE, "csrc", "build")
    os.makedirs(objdir, exist_ok=True)
    tag = os.path.splitext(os.path.basename(out))[0]
    base = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Rpass-analysis=kernel-resource-usage"] + list(extra_flags)
    jobs = []
    for src, flags in SOURCES.items():
        obj = os.path.join(objdir, f"{tag}.{os.path.splitext(src)[0]}.o")
        cmd = base + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    errs, objs = [], []
    for obj, proc in jobs:   # the units compile side by side
        _, err = proc.communicate()
        errs.append(err)
        objs.append(obj)
        if proc.returncode != 0:
            sys.stderr.write(err[-4000:])
            with open(log, "w") as f:
                f.write("".join(errs))
            raise RuntimeError(f"hipcc failed ({proc.returncode}) on {obj}; full log in {log}")
    with open(log, "w") as f:
        f.write("".join(errs))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise RuntimeError(f"link failed ({r.returncode})")
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
