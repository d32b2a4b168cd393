"""Memory-safety harness (GPU box): poisoned buffers and guard pages, one spawned child per (mode, variant).

Why: two of the three GPU faults of round 3 were out-of-bounds READS that surfaced only when a tensor happened to end an
allocator segment, and an 8-GPU launch changes every allocation layout ("Next round" 1).  tests/memsafety_child.py
runs the calls a trainer / inferencer makes through ONE engine -- a large batch, then a single mixture and a pair on the
same workspace, the stage entry points, a whole training step (tickets and static tile order), the path-level training
entry points --

  * "poison": on buffers that start as 0xFF bytes instead of zeros: results bit-identical to the zero-filled run, i.e. no
    kernel reads what this call has not written (stale workspace contents of another batch size included);
  * "guard_end" / "guard_start": with every weight tensor, input, output, workspace, tape and gradient buffer in an
    allocation of its own that ends / starts flush against an UNMAPPED page (HIP virtual-memory API, tests/guardmem): any
    out-of-bounds access is a GPU memory access fault.

A fault kills the child, not the suite; the failure message carries the child's last "BEGIN ..." line.
"""
import os
import subprocess
import sys

import pytest

from tests.memsafety_child import VARIANTS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("mode", ["poison", "guard_end", "guard_start"])
def test_memory_safety(mode, variant):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "tests.memsafety_child", mode, variant], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    out = r.stdout
    begins = [ln for ln in out.splitlines() if ln.startswith(("BEGIN", "=="))]
    last = " / ".join(begins[-2:]) if begins else "(nothing started)"
    assert r.returncode == 0, f"{mode} {variant}: child ended with code {r.returncode} during [{last}]\n{out[-3000:]}"
    assert f"OK {mode} {variant}" in out, out[-3000:]
