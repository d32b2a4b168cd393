#!/usr/bin/env python3
"""GPU box: device time of every kernel of ONE TransformerDPRNN alone on the chip (no second stream), fp32 default vs the
opt-in split-precision mode, at the bench shape.   python3 tools/path_iso.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_separation_amd.engine import DptnEngine, params_to_device  # noqa: E402
from speech_separation_amd.spec import DPTN_AV, synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = DPTN_AV
eng = DptnEngine(cfg, dev)
eng.bind(params_to_device(synthetic_state_dict(cfg, 0), dev))
S = eng.chunks(32000)
x = torch.randn(B, S, cfg.chunk_size, cfg.num_features, device=dev)
eng.set_option("overlap", 0)
for split in (0, 1):
    eng.set_option("split_bf16", split)
    for path in (0, 1):
        eng.stage_path(0, path, x)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(5):
            eng.stage_path(0, path, x)
        prof = eng.profile_read()
        eng.profile(False)
        t = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
        print(f"B={B} split={split} path={path}: total {sum(t.values()) * 1e3:8.1f} us   " + " ".join(f"{k}={v * 1e3:.0f}" for k, v in t.items()))
