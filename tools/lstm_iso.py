#!/usr/bin/env python3
"""Diagnostic (GPU box): per-kernel device time of ONE TransformerDPRNN run alone on the chip (no half-batch overlap),
for half / whole batches and both LSTM tile heights.   python3 tools/lstm_iso.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_state_dict  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
T = 32000
S = eng.chunks(T)
eng.profile(True)
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, S, cfg.chunk_size, cfg.num_features, device=dev)
    for tile in (4, 16, 32):
        eng.set_option("lstm16", 0 if tile == 32 else 1)
        eng.set_option("lstm4", 2 if tile == 4 else 0)
        for path, name in ((0, "intra"), (1, "inter")):
            eng.stage_path(0, path, x)
            eng.profile_reset()
            for _ in range(reps):
                eng.stage_path(0, path, x)
            prof = eng.profile_read()
            print(f"B={B:2d} tile={tile} {name}: " + "  ".join(f"{k}={v[0] / max(v[1], 1):.4f}" for k, v in prof.items() if v[1]))
