#!/usr/bin/env python3
"""Profiling driver (GPU box): run ONE TransformerDPRNN (or the whole forward) a few times at the bench shape so
that rocprofv3 --pmc passes stay short.   python3 tools/prof_path.py [intra|inter|forward] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_inputs, synthetic_state_dict  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "intra"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
B, T = 16, 32000
if what == "forward":
    inp = synthetic_inputs(cfg, B=B, T=T, Tv=50, seed=123)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    for _ in range(reps):
        eng.forward(t["mix"], t["s1_embedding"], t["s2_embedding"])
else:
    S = eng.chunks(T)
    x = torch.randn(B, S, cfg.chunk_size, cfg.num_features, device=dev)
    for _ in range(reps):
        eng.stage_path(0, 0 if what == "intra" else 1, x)
torch.cuda.synchronize()
print("done", what, reps)
