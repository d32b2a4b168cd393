"""TEST INFRASTRUCTURE: one memory-safety run of libdptnav in a process of its own (tests/test_gpu_memsafety.py spawns it;
a GPU memory access fault kills only this child, and its last "BEGIN ..." line names what was running).

    python -m tests.memsafety_child <mode> <variant>

mode  poison       every buffer the engine allocates (workspace, tapes, backward workspace, outputs, gradient buffers) starts
                   as 0xFF bytes (NaN as fp32, ~4e9 as a ticket counter) instead of zeros, and a large batch runs BEFORE a
                   small one on the same buffers: whatever a kernel reads without having written it this call shows up as a
                   difference to the zero-filled run.  Forward outputs, stage outputs and (option deterministic) parameter
                   gradients must be bit-identical.
      guard_end    weights (one allocation per tensor), inputs, outputs, workspaces, tapes and per-slot gradient buffers live
                   on guard pages (tests/guardmem): each ENDS flush against an unmapped page -> an out-of-bounds access
                   beyond a buffer is a GPU fault, not a silent read of the neighbouring tensor.  Results must equal the
                   plain run's.
      guard_start  the same with each buffer STARTING right behind an unmapped page (underruns).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from speech_separation_amd.engine import DptnEngine  # noqa: E402
from speech_separation_amd.spec import (DPRNN_AV, DPTN_AUDIO, DPTN_AV, DPTNConfig, synthetic_inputs,  # noqa: E402
                                        synthetic_state_dict)


def _cfg(base, **kw):
    return DPTNConfig(**{**base.to_dict(), "dropout": 0.0, **kw})


# name -> (config, options, big batch, T, Tv, train batch).  Small clips keep a child at a few seconds; "full" is the
# BASELINE configuration itself (B = 16 x 4 s, six blocks): the layout the 8-GPU run will see, with other addresses.
VARIANTS = {
    "dptn128": (_cfg(DPTN_AV, num_blocks=2), {}, 5, 8000, 13, 3),
    "dptn128_lstm32": (_cfg(DPTN_AV, num_blocks=2), {"lstm16": 0, "lstm4": 0}, 5, 8000, 13, 3),
    "dptn128_lstm4": (_cfg(DPTN_AV, num_blocks=2), {"lstm4": 2}, 5, 8000, 13, 3),
    "dptn128_lstm16x": (_cfg(DPTN_AV, num_blocks=2), {"lstm4": 0, "fuse_pre128": 2}, 5, 8000, 13, 0),   # input projection inside the recurrence
    "dptn128_pre": (_cfg(DPTN_AV, num_blocks=2), {"lstm4": 0, "fuse_pre128": 0}, 5, 8000, 13, 0),      # ... and the K4 + lstm16 path
    "dptn128_unfused": (_cfg(DPTN_AV, num_blocks=2), {"fuse_attn": 0, "fold_tail": 0, "pack_wih": 0, "pack_whh": 0}, 4, 6000, 7, 2),
    "dptn128_fc_engine": (_cfg(DPTN_AV, num_blocks=2), {"fcln": 0}, 4, 6000, 7, 2),    # training forward: out-projection / FFN + LayerNorm tape by the GEMM engine (default: fcln.hip)
    "dptn128_long": (_cfg(DPTN_AV, num_blocks=1), {}, 2, 48000, 50, 1),          # inter-chunk sequences > 160: streaming attention
    "dptn64": (_cfg(DPTN_AUDIO, num_blocks=2), {}, 5, 8000, 1, 3),
    "dptn64_lstm4": (_cfg(DPTN_AUDIO, num_blocks=2), {"lstm4": 2}, 3, 5000, 1, 2),
    "dptn64_lstm16x": (_cfg(DPTN_AUDIO, num_blocks=2), {"lstm4": 0}, 5, 8000, 1, 3),          # input projection inside the recurrence
    "dptn64_fc_engine": (_cfg(DPTN_AUDIO, num_blocks=2), {"fcln": 0}, 3, 5000, 1, 2),          # last FFN + LN2 / separation conv by the GEMM engine (default: fcln.hip)
    "dptn64_pre": (_cfg(DPTN_AUDIO, num_blocks=2), {"lstm4": 0, "fuse_pre": 0}, 3, 5000, 1, 0),  # ... and the K4 + lstm16 path
    "dprnn": (_cfg(DPRNN_AV, num_blocks=2), {}, 3, 4000, 9, 2),
    "dprnn_lstm32": (_cfg(DPRNN_AV, num_blocks=2), {"lstm16": 0, "lstm4": 0}, 3, 4000, 9, 2),
    "dprnn_lstm16x": (_cfg(DPRNN_AV, num_blocks=2), {"lstm4": 0}, 3, 4000, 9, 0),
    "dprnn_fcln3": (_cfg(DPRNN_AV, num_blocks=2), {"fcln": 2}, 3, 4000, 9, 0),              # fcln.hip, two tiles ahead (default: one)
    "dprnn_fc_engine": (_cfg(DPRNN_AV, num_blocks=2), {"fcln": 0}, 3, 4000, 9, 0),          # ... and the GEMM engine's fc + LayerNorm
    "unidir128": (_cfg(DPTN_AV, num_blocks=2, bidir=False), {}, 5, 8000, 13, 3),
    "unidir64": (_cfg(DPTN_AUDIO, num_blocks=2, bidir=False), {}, 4, 8000, 1, 2),
    "split_bf16": (_cfg(DPTN_AV, num_blocks=2), {"split_bf16": 1}, 5, 8000, 13, 0),
    "ragged": (_cfg(DPTN_AV, num_blocks=1, chunk_size=50, step_size=25, kernel_size_enc=5), {}, 3, 3001, 5, 2),
    "full": (_cfg(DPTN_AV), {}, 16, 32000, 50, 16),
}


def say(msg):
    print(msg, flush=True)


def run_sequence(eng, cfg, inputs_of, Bbig, T, Tv, Btrain, place, full):
    """The calls a trainer / inferencer makes through one engine, big batch first.  -> {name: numpy array}."""
    res = {}

    def fwd(tag, B):
        say(f"BEGIN forward B={B}")
        t = inputs_of(B)
        s1, s2 = eng.forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
        torch.cuda.synchronize()
        res[f"{tag}.s1"], res[f"{tag}.s2"] = s1.cpu().numpy(), s2.cpu().numpy()

    fwd("fwd_big", Bbig)
    fwd("fwd_one", 1)                       # the small batch runs on what the big one left in the workspace
    if not full:
        fwd("fwd_two", 2)
        say("BEGIN stage entry points")
        t = inputs_of(2)
        enc, chk = eng.stage_head(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
        x = chk
        for block in range(cfg.num_blocks):
            for path in (0, 1):
                x = eng.stage_path(block, path, x)
        s1, s2 = eng.stage_tail(x, enc, T)
        torch.cuda.synchronize()
        res["stage.enc"], res["stage.x"], res["stage.s1"], res["stage.s2"] = (a.cpu().numpy() for a in (enc, x, s1, s2))
    if Btrain:
        for det in ((1,) if full else (1, 0)):
            eng.set_option("deterministic", det)
            say(f"BEGIN training step B={Btrain} deterministic={det}")
            t = inputs_of(Btrain)
            rng = np.random.default_rng(11)
            d1 = place(torch.from_numpy(rng.standard_normal((Btrain, T)).astype(np.float32)))
            d2 = place(torch.from_numpy(rng.standard_normal((Btrain, T)).astype(np.float32)))
            s1, s2, tape = eng.train_forward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"))
            eng.train_backward(t["mix"], t.get("s1_embedding"), t.get("s2_embedding"), d1, d2, tape)
            torch.cuda.synchronize()
            res[f"train{det}.s1"] = s1.cpu().numpy()
            for k, g in eng._grads.items():
                res[f"train{det}.grad.{k}"] = g.cpu().numpy()
            del tape
        eng.set_option("deterministic", 0)
        if not full:
            say("BEGIN path-level training entry points")
            eng.set_option("deterministic", 1)
            S = eng.chunks(T)
            rng = np.random.default_rng(12)
            x = place(torch.from_numpy(rng.standard_normal((2, S, cfg.chunk_size, cfg.num_features)).astype(np.float32)))
            dy = place(torch.from_numpy(rng.standard_normal((2, S, cfg.chunk_size, cfg.num_features)).astype(np.float32)))
            for path in (0, 1):
                y, tape = eng.train_path_forward(0, path, x)
                dx = eng.train_path_backward(0, path, x, dy, tape)
                torch.cuda.synchronize()
                res[f"path{path}.y"], res[f"path{path}.dx"] = y.cpu().numpy(), dx.cpu().numpy()
            eng.set_option("deterministic", 0)
    return res


def main(mode, variant):
    cfg, options, Bbig, T, Tv, Btrain = VARIANTS[variant]
    full = variant == "full"
    dev = torch.device("cuda:0")
    sd = synthetic_state_dict(cfg, seed=5)
    host_inputs = {B: synthetic_inputs(cfg, B=B, T=T, Tv=Tv, seed=77 + B) for B in {Bbig, 1, 2, max(Btrain, 1)}}

    def make(alloc, place, per_slot):
        eng = DptnEngine(cfg, dev, alloc=alloc)
        params = {k: place(torch.from_numpy(v)) for k, v in sd.items()}
        eng.bind(params)
        if Btrain:
            eng.bind_grads(per_slot=per_slot)
        for k, v in options.items():
            eng.set_option(k, v)

        def inputs_of(B):
            return {k: place(torch.from_numpy(v)) for k, v in host_inputs[B].items() if k in ("mix", "s1_embedding", "s2_embedding")}
        return eng, inputs_of

    def plain_run():      # zero-filled buffers from the caching allocator
        say(f"== {mode} {variant}: plain run")
        eng, inputs_of = make(lambda n: torch.zeros(n, dtype=torch.uint8, device=dev), lambda t: t.to(dev), False)
        res = run_sequence(eng, cfg, inputs_of, Bbig, T, Tv, Btrain, lambda t: t.to(dev), full)
        eng.close()
        del eng, inputs_of
        torch.cuda.empty_cache()
        return res

    def run_under_test(m):
        say(f"== {m} {variant}: run under test")
        arena = None
        if m == "poison":
            eng, inputs_of = make(lambda n: torch.full((n,), 0xFF, dtype=torch.uint8, device=dev), lambda t: t.to(dev), False)
            place = lambda t: t.to(dev)   # noqa: E731
        else:
            from tests.guardmem import GuardArena
            arena = GuardArena(0, flush="end" if m == "guard_end" else "start", fill=0xFF)
            place = lambda t: arena.like(t.contiguous())   # noqa: E731
            eng, inputs_of = make(lambda n: arena.bytes(n, 256), place, True)
        res = run_sequence(eng, cfg, inputs_of, Bbig, T, Tv, Btrain, place, full)
        eng.close()
        del eng, inputs_of, place
        if arena is not None:
            ch = arena.chunks()
            say(f"guard arena: {len(arena.handles)} allocations, {arena.total / 2**20:.1f} MiB, granularity {arena.granularity}, "
                f"largest backed by {max(ch)} physical handles")
            arena.close()           # every unmap / release / address-free return code is checked
        torch.cuda.empty_cache()
        return res

    if mode == "all":
        # ONE-OFF: the history that preceded both faults of round 4 -- plain, poison and guard_end in ONE
        # process, in that order -- against the hardened allocator.  Not part of the suite.
        want = plain_run()
        runs = [(m, run_under_test(m)) for m in ("poison", "guard_end")]
    else:
        # the run under test FIRST: a guard arena then reserves its ranges in a process that has not yet handed 43 GB tapes back
        # to the driver (the per-mode children still ran the plain pass in front of the arena)
        got = run_under_test(mode)
        want = plain_run()
        runs = [(mode, got)]

    # ---- compare
    rc = 0
    for m, got in runs:
        bad = []
        for k in want:
            a, b = want[k], got[k]
            if not np.all(np.isfinite(b)):
                bad.append(f"{k}: non-finite values ({int(np.sum(~np.isfinite(b)))} of {b.size})")
            elif k.startswith("train0.grad."):      # ticket order varies between runs: fp32 summation order only
                den = float(np.abs(a).max()) or 1.0
                if float(np.abs(a - b).max()) > 1e-4 * den:
                    bad.append(f"{k}: differs by {float(np.abs(a - b).max()) / den:.2e} of its maximum")
            elif not np.array_equal(a, b):
                bad.append(f"{k}: not bit-identical (max |d| {float(np.abs(a - b).max()):.3e}, {int(np.sum(a != b))} of {a.size} elements)")
        if bad:
            say(f"FAILED {m}\n  " + "\n  ".join(bad[:40]))
            rc = 1
        else:
            say(f"OK {m} {variant}: {len(want)} results identical")
    return rc


if __name__ == "__main__":
    rc = main(sys.argv[1], sys.argv[2])
    sys.stdout.flush()
    os._exit(rc)      # no interpreter teardown with guard mappings still referenced by tensors
