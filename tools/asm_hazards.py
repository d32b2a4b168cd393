#!/usr/bin/env python3
"""Hazards the compiler's recogniser cannot see because the instructions sit in inline assembly -- checked in the GENERATED
code (hipcc cross-compiles gfx950 without a GPU):
  * an LDS-DMA request (global_load_lds_*) reads M0: the ISA asks for one wait state behind the scalar write of M0;
  * a store of more than 8 bytes reads its data registers for a few cycles after issue: the next instruction must not be a
    vector write (round 5, dgrad_t.hip: without the s_nop lanes 8-15 / 24-31 of each half stored the NEXT store's values);
  * a vector-memory instruction that takes a SCALAR operand (base pointer) needs five wait states behind a VALU write of that
    register (v_readlane of a spilled SGPR, v_readfirstlane): the statement cannot know what the allocator put in front of it, so
    every such statement starts with s_nop 4 (round 5, dgrad_r.hip: a request with a stale base = memory access fault);
  * dgrad_t.hip loads its weights into AGPRs by loads the compiler does not count: a register copy (v_accvgpr_*) or a spill
    (scratch_*) anywhere in the kernel could read them before they have arrived.
    python3 tools/asm_hazards.py        exit code 1 and a list when something is found"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech_separation_amd", "csrc")
FILES = ["dgrad_t.hip", "dgrad_r.hip", "gemm_t.hip", "attn_block2.hip", "lstm16x.hip", "fcln.hip"]
NO_VGPR_FORM = {"dgrad_r.hip"}


def asm_of(src):
    out = os.path.join("/tmp", "asm_hazards_" + os.path.splitext(src)[0] + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + ([] if src in NO_VGPR_FORM else ["-mllvm", "-amdgpu-mfma-vgpr-form"]) + [
        "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"{src}: hipcc failed\n{r.stderr[-2000:]}")
    return [ln.strip() for ln in open(out) if ln.strip() and not ln.strip().startswith(";") or ln.strip().startswith(";;#")]


def main():
    bad = []
    for src in FILES:
        lines = asm_of(src)
        n_dma = n_store = n_sbase = 0
        in_asm = False
        block_start = 0
        func = "?"
        for i, ln in enumerate(lines):
            if ln.endswith(":") and ln.startswith("_Z"):
                func = ln[:60]
            if ln.startswith(";;#ASMSTART"):
                in_asm = True
                block_start = i
                continue
            if ln.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not in_asm:
                continue
            if re.match(r"global_(load|store|atomic)", ln) and re.search(r"s\[\d+:\d+\]", ln):
                n_sbase += 1
                nops = [int(p.split()[1]) for p in lines[block_start + 1:i] if p.startswith("s_nop")]
                if not nops or max(nops) < 4:
                    bad.append(f"{src} {func}: vector-memory instruction with a scalar base and no s_nop 4 in front of it in its statement: {ln}")
            if ln.startswith("global_load_lds"):
                n_dma += 1
                prev = lines[block_start + 1:i]      # the statement's own instructions in front of the request
                k = max((j for j, p in enumerate(prev) if p.startswith("s_mov_b32 m0")), default=None)
                if k is None or not any(not p.startswith("s_mov_b32 m0") for p in prev[k + 1:]):
                    bad.append(f"{src} {func}: LDS-DMA directly behind the write of M0: {prev} -> {ln}")
            if re.match(r"global_store_dwordx[34]", ln):
                n_store += 1
                if not lines[i + 1].startswith("s_nop"):
                    bad.append(f"{src} {func}: wide store in inline assembly without a wait state behind it: {ln} / {lines[i + 1]}")
        if src in ("dgrad_t.hip", "gemm_t.hip"):
            body = "\n".join(lines)
            for pat in ("v_accvgpr_", "scratch_"):
                if pat in body:
                    bad.append(f"{src}: {body.count(pat)} x {pat} (weights loaded by uncounted loads must not be copied or spilled)")
        print(f"{src}: {n_dma} hand-issued LDS-DMA requests, {n_store} hand-issued wide stores, {n_sbase} scalar-base operands checked")
    for b in bad:
        print("HAZARD:", b)
    print(f"{len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
