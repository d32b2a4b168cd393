#!/usr/bin/env python3
"""GPU box: forward time over (batch, sub-batches, recurrence launches in flight) for the two DPTN configurations -- the
data behind forward_split's rule (options sub_batches / lstm_inflight).   python3 tools/split_sweep.py [lstm4 [B ...]]
(lstm4: 0 / 1 / 2 = the 4-sequence recurrence never / where one round fits / wherever the 16-tile layout is in use)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AUDIO, DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
LSTM4 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BATCHES = [int(a) for a in sys.argv[2:]] or [4, 8, 12, 16, 20, 24, 32]
print(f"lstm4 = {LSTM4}")
for name, cfg in (("dptn_av", DPTN_AV), ("dptn_audio", DPTN_AUDIO)):
    eng = DptnEngine(cfg, dev)
    eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
    eng.set_option("lstm4", LSTM4)
    for kv in filter(None, os.environ.get("SWEEP_OPTS", "").split(",")):      # e.g. SWEEP_OPTS=fuse_pre128=1,fuse_pre=0
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for B in BATCHES:
        inp = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_inputs(cfg, B=B, T=32000, Tv=50, seed=0).items()}
        args = (inp["mix"], inp.get("s1_embedding"), inp.get("s2_embedding"))
        res = []
        for nsub, depth in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (8, 0)):
            if nsub > B:
                continue
            eng.set_option("sub_batches", nsub)
            eng.set_option("lstm_inflight", depth)
            try:
                for _ in range(2):
                    eng.forward(*args)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    eng.forward(*args)
                e1.record()
                torch.cuda.synchronize()
                res.append(f"{nsub}/{depth}: {e0.elapsed_time(e1) / 6:6.2f}")
            except RuntimeError as e:
                res.append(f"{nsub}/{depth}: err")
        print(f"{name} B={B:2d}  " + "  ".join(res), flush=True)
    del eng
    torch.cuda.empty_cache()
