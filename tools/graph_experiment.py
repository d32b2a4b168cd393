#!/usr/bin/env python3
"""GPU box, EXPERIMENT: the forward as a captured hipGraph (engine.ForwardGraph) against eager launches, every call
synchronised (serving) -- DESIGN.md section 6.4.  Inputs are generated on the device and kept alive (a GPU memory fault was
seen at B = 1 when host buffers were freed between replays).   python3 tools/graph_experiment.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
keep = []
for B in (1, 4, 16):
    mix = 0.1 * torch.randn(B, 32000, device=dev)
    e1, e2 = torch.randn(B, 512, 50, device=dev), torch.randn(B, 512, 50, device=dev)
    out = (torch.empty_like(mix), torch.empty_like(mix))
    keep += [mix, e1, e2, out]

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return sum(ts) / n, min(ts)
    em, en = timeit(lambda: eng.forward(mix, e1, e2, out=out))
    ref = out[0].clone()
    g = eng.capture_forward(B, 32000, 50)
    keep.append(g)
    g(mix, e1, e2)
    torch.cuda.synchronize()
    same = torch.equal(g.out[0], ref)
    gm, gn = timeit(g.replay)
    print(f"B={B}: eager {em:.3f} ms (min {en:.3f}), hipGraph replay {gm:.3f} ms (min {gn:.3f}), bit-identical {same}", flush=True)
