#!/usr/bin/env python3
"""GPU box, MEASUREMENT ONLY: how much the training FORWARD would gain if its attention half (QKV GEMM +
attention + out-projection with LayerNorm tape) were the fused inference block -- option train_fuse_probe runs exactly that
(the block writes y1 only: no qkv / att / LayerNorm tape, so a backward after it would be garbage; none is run).  The gain
measured here is the UPPER bound of a fused front half that also stores what the backward needs.
    python3 tools/train_fuse_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
eng = DptnEngine(DPTN_AV, dev)
eng.bind(params_to_device(synthetic_state_dict(DPTN_AV, 0), dev))
eng.bind_grads()
t = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(DPTN_AV, B=16, T=32000, Tv=50, seed=1).items()}
args = (t["mix"], t["s1_embedding"], t["s2_embedding"])
res = {0: [], 1: []}
for r in range(4):
    for v in (0, 1):
        eng.set_option("train_fuse_probe", v)
        for _ in range(2):
            s1, s2, tape = eng.train_forward(*args)
            del tape
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            s1, s2, tape = eng.train_forward(*args)
            del tape
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / 8)
for v in res:
    print(f"train_fuse_probe={v}: training forward {np.mean(res[v]):.3f} ms ({' '.join(f'{x:.3f}' for x in res[v])})")
print(f"upper bound of a fused attention front half in the training forward: {np.mean(res[0]) - np.mean(res[1]):.3f} ms per step")
