"""Checks on the GENERATED code of the HIP library (no GPU needed: hipcc cross-compiles gfx950 here)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_tile_kernel_waits_for_its_ticket():
    """the GEMM engine / weight-gradient kernels request the ticket of the tile after the next one in front of
    their MFMA block; if the register allocator parks the atomic's result in an AGPR it has to wait for it on the spot (the
    A-tile prefetch's HBM latency + the atomic's round trip on every tile: 9 % of the K4 kernel when it happened).
    tools/ticket_waits.py compiles csrc/dptnav.hip to assembly and fails if any of the 88 instantiations has a wait in the
    basic block of its loop atomic."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ticket_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 wait for the one inside their tile loop" in r.stdout


def test_inline_assembly_memory_operations_carry_their_wait_states():
    """Round 5: instructions issued from inline assembly are invisible to the compiler's hazard recogniser.  tools/asm_hazards.py
    compiles the kernels that issue LDS-DMA requests / wide stores by hand and checks the generated code: a wait state between the
    scalar write of M0 and the LDS-DMA that reads it, one behind every 16-byte store (dgrad_t.hip stored the next store's values
    from half of its lanes without it), and no register copy or spill in the kernel whose weights arrive by uncounted loads."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazards.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 hazards" in r.stdout
