#!/usr/bin/env python3
"""Phase stamps of the fused attention block (attn_block.hip built with -DATTN_STAMPS into its own library):
cycles per sequence and wave spent in each phase of the kernel, as run inside dptnav_forward (B = 16, two streams) and
with the sub-batches one after the other (option overlap = 0).

    python3 tools/attn_stamps.py build      # here (hipcc cross-compiles): speech_separation_amd/libdptnav_attnstamps.so
    python3 tools/attn_stamps.py            # on the GPU box
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "speech_separation_amd", "libdptnav_attnstamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from speech_separation_amd.build import build_lib
    print(build_lib(force=True, out=DIAG, extra_flags=["-DATTN_STAMPS"]))
    sys.exit(0)

import torch  # noqa: E402
import speech_separation_amd._lib as L  # noqa: E402
L.LIB_PATH = DIAG
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

lib = L.load()
lib.dptnav_debug_attn_stamps.argtypes = [C.c_void_p, C.c_int]
lib.dptnav_debug_attn2_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=16, T=32000, Tv=50, seed=0).items()}
args = (inp["mix"], inp["s1_embedding"], inp["s2_embedding"])
NAMES = {0: ["PRO: first fetch", "PRO: stage h rows + barrier", "PRO: 128 MFMAs per block", "PRO: product -> LDS + barrier", "PRO: LayerNorm 2 rows",
             "K/V weights (+ x staging) + barrier", "phase 1: K^T / V tiles", "phase-2 constants + first Q tile", "softmax loop",
             "barrier + out-projection + barrier", "LayerNorm 1 rows beside the next Q tile"],
         # attn_block2.hip: the marks sit at the same places of the new structure
         1: ["PRO: first fetch", "PRO: wait for the block's DMA + barrier", "PRO: 128 MFMAs per block", "PRO: + bias + residual, group statistics, exchange barrier",
             "PRO: merge, normalise, x -> LDS (+ next DMA issue)", "K/V weights (+ x staging) + barrier", "phase 1: K^T / V tiles", "phase-2 constants + first Q tile",
             "softmax loop", "O^T exchange + out-projection + statistics + barrier", "normalise + store beside the next Q tile"]}
ideal = [0, 0, 5 * 128 * 64, 0, 0, 0, 5 * 128 * 64, 64 * 64, 5 * 160 * 64, 5 * 64 * 64, 4 * 64 * 64]
variants = [int(a) for a in sys.argv[1:]] or [1, 0]
for v2 in variants:
    eng.set_option("attn_v2", v2)
    fn = lib.dptnav_debug_attn2_stamps if v2 else lib.dptnav_debug_attn_stamps
    for overlap in (1, 0):
        eng.set_option("overlap", overlap)
        for _ in range(2):
            eng.forward(*args)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        fn(buf, 1)
        eng.forward(*args)
        fn(buf, 0)
        v = list(buf)
        n = max(v[12], 1)
        print(f"attn_v2={v2} ({'attn_block2.hip' if v2 else 'attn_block.hip'}) overlap={overlap}: {n} wave-sequences; cycles per sequence and wave "
              f"(MFMA issue cycles of the phase in brackets)")
        for k, nm in enumerate(NAMES[v2]):
            print(f"  {nm:62s} {v[k] / n:9.0f}  [{ideal[k]}]")
        print(f"  {'total':62s} {sum(v[:11]) / n:9.0f}  [{sum(ideal)}]")
