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
SOURCES = ["dptnav.hip"]
HEADERS = ["common.h", "gemm_ws.h", "attention.h", "lstm.h", "headtail.h", os.path.join("..", "..", "include", "dptnav.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    log = os.path.join(HERE, "csrc", "build_resource_usage.log")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    with open(log, "w") as f:
        f.write(r.stderr)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise RuntimeError(f"hipcc failed ({r.returncode}); full log in {log}")
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
