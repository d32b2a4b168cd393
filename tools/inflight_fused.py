#!/usr/bin/env python3
"""GPU box: forward time at B = 16 over (sub-batches, recurrence launches in flight) with the input projection inside the
recurrence (fuse_pre128 / fuse_pre), against the default schedule.   python3 tools/inflight_fused.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg, B = DPTN_AV, 16
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=0).items()}
args = (inp["mix"], inp["s1_embedding"], inp["s2_embedding"])


def timed(n=10):
    for _ in range(3):
        eng.forward(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng.forward(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for fused in (0, 1):
    eng.set_option("fuse_pre128", fused)
    row = []
    for nsub, depth in ((0, 0), (3, 2), (3, 1), (4, 2), (4, 3), (2, 1)):
        eng.set_option("sub_batches", nsub)
        eng.set_option("lstm_inflight", depth)
        row.append(f"{nsub}/{depth}: {timed():6.2f}")
    print(f"fuse_pre128={fused}  " + "  ".join(row), flush=True)
